"""In-kernel timeline of big_chol_dataflow on one dense supernode (diagnostic build: make -C
conex_amd/csrc dbg).  Waves 0 (diagonal block) and 1 (first 64 rows below it) of every workgroup stamp
s_memrealtime (100 MHz, chip-wide).  Run on the GPU box:  python tools/big_chol_stamps.py [n]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import conex_amd.kkt as kk

kk.LIB_PATH = os.path.join(os.path.dirname(kk.LIB_PATH), os.environ.get("CXK_DBG_LIB", "libconex_dbg.so"))
from conex_amd import KktContext, synthetic as syn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
L = kk.load_library()
prob = syn.lp_problem(rows=n + 100, num_vars=n, seed=3)
ctx = syn.build(KktContext, prob, "lp", device=0)
ctx.set_cost(prob["b"])
for _ in range(10):
    ctx.kkt_solve_async(0.7, 0.9, 0.8)
assert ctx.sync()
nb = (n + 31) // 32
L.cxk_debug_big_chol_stamps.argtypes = [C.POINTER(C.c_longlong), C.c_int]
buf = (C.c_longlong * (16 * nb))()
assert L.cxk_debug_big_chol_stamps(buf, nb) == 0
s = np.array(buf[:], dtype=np.int64).reshape(nb, 2, 8)
t0 = s[:, :, 0].min()
us = (s - t0) / 100.0
print("block | wave 0: entry  wait(j-1)  flag seen  updated  to rows  eliminated  (A)  (B) | wave 1: entry  wait(j-1)  flag B seen  updated  (A)  solved  flag A set")
for j in range(nb):
    a, b = us[j, 0], us[j, 1]
    print("%5d | " % j + " ".join("%7.2f" % v for v in a) + " | " + " ".join("%7.2f" % v for v in b[:7]))
