import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import conex_amd.kkt as kk
kk.LIB_PATH = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scratch", "libconex_dbg.so")
from conex_amd import KktContext
from conex_amd import synthetic as syn
prob = syn.lmi_problem(); W = syn.scaling_points(1000, 20)
k = syn.build(KktContext, prob, "lmi", device=0)
for i in range(k.K): k.set_W(i, W[i])
k.set_cost(prob["b"])
for _ in range(5): k.kkt_solve_async(0.7,0.9,0.8)
k.sync()
L = kk.load_library()
st=(C.c_longlong*64)(); L.cxk_debug_fused_stamps(st)
v=np.array(list(st)).reshape(8,8)
for w in range(7):
    print("wave",w,"deltas",[int(v[w,i+1]-v[w,i]) for i in range(6)], "total", int(v[w,6]-v[w,0]))
