"""In-kernel timeline of lmi_schur_mfma on the C4 workload (diagnostic build: make -C conex_amd/csrc dbg).

Prints, for workgroups 0 and 200, the s_memtime stamps of every wave relative to the
workgroup's first stamp (shader cycles): producers (waves 0-3) and consumers (waves 4-7).
Run on the GPU box:  python tools/mfma_stamps.py [K] [m]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import conex_amd.kkt as kk

kk.LIB_PATH = os.path.join(os.path.dirname(kk.LIB_PATH), os.environ.get("CXK_DBG_LIB", "libconex_dbg.so"))
from conex_amd import KktContext, synthetic as syn

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L = kk.load_library()
prob = syn.lmi_problem(K=K, n=20, m=m, branching=8, overlap=min(5, m - 1))
W = syn.scaling_points(K, 20)
ctx = syn.build(KktContext, prob, "lmi", device=0)
for i in range(ctx.K):
    ctx.set_W(i, W[i])
ctx.set_cost(prob["b"])
for _ in range(10):
    ctx.kkt_solve_async(0.7, 0.9, 0.8)
ctx.sync()
L.cxk_debug_mfma_stamps.argtypes = [C.POINTER(C.c_longlong)]
buf = (C.c_longlong * (2 * 16 * 64))()
assert L.cxk_debug_mfma_stamps(buf) == 0
s = np.array(buf[:], dtype=np.int64).reshape(2, 16, 64)
for b, name in enumerate(("workgroup 0", "workgroup 200")):
    t0 = s[b][s[b] > 0].min()
    print(name, "(cycles since the workgroup's first stamp; per iteration: start / tiles or contraction done / results written)")
    for w in range(12):
        row = s[b, w]
        role = "producer" if w < 8 else "consumer"
        its = []
        for it in range(0, 8):
            q = row[1 + 4 * it:5 + 4 * it]
            if q[0] <= 0:
                continue
            its.append("it%d[%s]" % (it, " ".join(str(int(v - t0)) if v > 0 else "-" for v in (q[0], q[1], q[3]))))
        fine = [int(v - t0) for v in row[48:60] if v > 0]
        print("  wave %2d %s: first %5d | %s%s" % (w, role, row[0] - t0, " ".join(its), (" | iteration 2, per tile slot (start, MFMAs issued): %s" % fine) if fine else ""))
    print("  span %d cycles" % (s[b].max() - t0))
