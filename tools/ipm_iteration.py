"""Wall time per interior-point iteration of CONEX_Maximize on the C4 program (BASELINE config 4)
through the conex.h C-ABI, and the device time of the reference's phases (CONEX_ENABLE_TIMER).
Run on the GPU box:  python tools/ipm_iteration.py [--timers]
Under rocprofv3 (tools/ipm_rocprof.sh) the kernel trace gives the per-kernel shares of an iteration.
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if "--timers" in sys.argv:
    os.environ["CONEX_ENABLE_TIMER"] = "1"
from conex_amd import capi as ca
from conex_amd import synthetic as syn

def _opt(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


# --K / --n / --m: another shape of the same chordal program (orders other than 20 take the
# workgroup LMI kernels: no tail workgroup, the selection of mu rides in the reduction launch)
prob = syn.lmi_problem(K=_opt("--K", 1000), n=_opt("--n", 20), m=_opt("--m", 20), branching=8,
                       overlap=min(5, _opt("--m", 20) - 1))
L = ca.api()
n = prob["n"]


def build():
    p = L.CONEX_CreateConeProgram()
    assert L.CONEX_SetNumberOfVariables(p, prob["num_vars"]) == 0
    for c, cl in enumerate(prob["cliques"]):
        a, cm = ca.colmajor(prob["A"][c]), ca.colmajor(prob["C"][c])
        v = np.ascontiguousarray(cl, dtype=np.int64)
        assert L.CONEX_AddSparseLMIConstraint(p, ca.dp(a), n, n, len(cl), ca.dp(cm), n, n,
                                              v.ctypes.data_as(C.POINTER(C.c_long)), len(cl)) == c
    return p


b = np.ascontiguousarray(prob["b"], dtype=np.float64)
p = build()
for rep in range(3):
    cfg = ca.default_config()
    y = np.zeros(len(b))
    t0 = time.perf_counter()
    ok = L.CONEX_Maximize(p, ca.dp(b), len(b), C.byref(cfg), ca.dp(y), len(b))
    dt = time.perf_counter() - t0
    st = ca.IterationStats()
    L.CONEX_GetIterationStats(p, C.byref(st), -1)
    iters = st.iteration_number + 1
    print("solve %d: ok=%d iterations=%d wall %.2f ms = %.1f us per iteration (first solve includes set-up)" % (rep, ok, iters, dt * 1e3, dt * 1e6 / iters), flush=True)
if "--timers" in sys.argv:
    L.CONEX_HIP_GetPhaseTimes.argtypes = [C.c_void_p, ca.c_double_p]
    us = np.zeros(5)
    L.CONEX_HIP_GetPhaseTimes(p, ca.dp(us))
    print("phase device time of the last solve (us): sparsity %.0f assemble %.0f factor %.0f solve %.0f update %.0f" % tuple(us))
L.CONEX_DeleteConeProgram(p)
