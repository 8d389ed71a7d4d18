"""In-kernel timeline of tree_fused on the C4 workload (diagnostic build: make -C conex_amd/csrc dbg).

Every wavefront (= supernode) stamps s_memrealtime (100 MHz, chip-wide) at: 0 entry, 1 record
decoded, 2 own panel assembled, 3 descendants' values in, 4 eliminated, 5 separator's solution in,
6 done.  Printed per tree level: median / max of each stamp in microseconds since the first entry.
Run on the GPU box:  python tools/fused_tree_stamps.py [K [n [m]]]
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
n_ = int(sys.argv[2]) if len(sys.argv) > 2 else 20      # order of the LMIs
m_ = int(sys.argv[3]) if len(sys.argv) > 3 else 20      # variables per LMI (python tools/fused_tree_stamps.py 1 200 50: config 2)
L = kk.load_library()
prob = syn.lmi_problem(K=K, n=n_, m=m_, branching=8 if K > 1 else 2, overlap=5 if K > 1 else 1)
W = syn.scaling_points(K, n_)
ctx = syn.build(KktContext, prob, "lmi", device=0)
assert ctx.fused_tree()
for i in range(ctx.K):
    ctx.set_W(i, W[i])
ctx.set_cost(prob["b"])
for _ in range(30):
    ctx.kkt_solve_async(0.7, 0.9, 0.8)
assert ctx.sync()
L.cxk_debug_fused_tree_stamps.argtypes = [C.POINTER(C.c_longlong), C.c_int]
buf = (C.c_longlong * (16 * K))()
assert L.cxk_debug_fused_tree_stamps(buf, K) == 0
s = np.array(buf[:], dtype=np.int64).reshape(K, 16)
t0 = s[:, 0].min()
lev = s[:, 15]
order = [0, 1, 2, 3, 4, 7, 8, 9, 5, 6]
us = (s[:, order] - t0) / 100.0
names = ["entry", "record", "assembled", "pulls in", "eliminated", "published", "L stored", "L columns", "y(sep) in", "done"]
print("level  count | " + " | ".join("%-13s" % n for n in names) + "   (median / max, us since the first entry)")
for l in sorted(set(lev.tolist())):
    sel = us[lev == l]
    print("%5d %6d | " % (l, len(sel)) + " | ".join("%5.2f / %5.2f" % (np.median(sel[:, i]), sel[:, i].max()) for i in range(len(order))))
    print("             polls until the descendants' values were in: median %d max %d; until the separator's solution was in: median %d max %d"
          % (np.median(s[lev == l, 12]), s[lev == l, 12].max(), np.median(s[lev == l, 14]), s[lev == l, 14].max()))
print("span %.2f us" % us[:, -1].max())

# placement: how many wavefronts of the launch share a SIMD (HW_ID: SIMD bits 5:4, CU 11:8, SH 12, SE 15:13; XCC_ID 3:0)
hw, xcc = s[:, 10], s[:, 11] & 15
simd = ((xcc << 16) | (hw & 0xFF00) | ((hw >> 4) & 3)).astype(np.int64)
cu = ((xcc << 16) | (hw & 0xFF00)).astype(np.int64)
_, inv, cnt = np.unique(simd, return_inverse=True, return_counts=True)
share = cnt[inv]
print("wavefronts per occupied SIMD: " + ", ".join("%d x%d" % (c, (cnt == c).sum()) for c in sorted(set(cnt.tolist()))),
      "; CUs occupied %d, XCDs %d" % (len(set(cu.tolist())), len(set(xcc.tolist()))))
el = (s[:, 4] - s[:, 3]) / 100.0
for l in sorted(set(lev.tolist()))[:2]:
    for c in sorted(set(share[lev == l].tolist())):
        sel = (lev == l) & (share == c)
        print("  level %d, %d on the SIMD: %4d wavefronts, elimination median %.2f max %.2f us, eliminated at median %.2f max %.2f"
              % (l, c, sel.sum(), np.median(el[sel]), el[sel].max(), np.median(us[sel, 4]), us[sel, 4].max()))
