# Repeats the interior-point solve of tests/test_gpu_solver.py::test_lp_dense_optimality_and_dual_recovery
# from process start and counts results that differ in any bit from the first:
#   python tools/det_lp.py 600        (CXK_NO_SPIN=1 replaces the mailbox spin-wait by a stream synchronize)
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conex_amd import capi as ca
import oracle_lib as ol
import test_gpu_solver as tgs
L = ca.api()
rng = np.random.default_rng(1)
cfg = ca.default_config()
cfg.prepare_dual_variables = 1
cfg.inv_sqrt_mu_max = 5e5
cfg.divergence_upper_bound = 1000
cfg.dinf_upper_bound = 1.35
cfg.final_centering_tolerance = 1
probs = []
for i in range(6):
    nv, nc = 5, 6 + 2 * i
    A = rng.uniform(-1, 1, (nc, nv)); c = np.abs(rng.uniform(-1, 1, nc))
    x0 = np.abs(rng.uniform(-1, 1, nc)); x0 *= 0.01 / np.linalg.norm(x0)
    probs.append((A, c, A.T @ x0, nv, nc))
ref_h, ref_o = {}, {}
bad_h = bad_o = 0
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for rep in range(REPS):
    for k, (A, c, b, nv, nc) in enumerate(probs):
        p = L.CONEX_CreateConeProgram()
        assert L.CONEX_AddDenseLinearConstraint(p, ca.dp(ca.colmajor(A)), nc, nv, ca.dp(c), nc) == 0
        ok, y = tgs._maximize(L, p, b, cfg)
        st = ca.IterationStats(); L.CONEX_GetIterationStats(p, C.byref(st), -1)
        L.CONEX_DeleteConeProgram(p)
        key = (ok, st.iteration_number, y.tobytes())
        if k not in ref_h: ref_h[k] = key
        elif ref_h[k] != key:
            bad_h += 1
            print("HIP differs: rep", rep, "problem", k, "iters", st.iteration_number, "vs", ref_h[k][1], "max |dy|", np.abs(y - np.frombuffer(ref_h[k][2])).max(), flush=True)
        if rep < 30:
            o = ol.Program(nv); o.add_linear(A, c)
            oko, yo = o.solve(b, tgs._sync_cfg(cfg))
            ko = (oko, yo.tobytes())
            if k not in ref_o: ref_o[k] = ko
            elif ref_o[k] != ko:
                bad_o += 1
                print("ORACLE differs: rep", rep, "problem", k, flush=True)
print("HIP nondeterministic results:", bad_h, "of", REPS * 6, "; oracle:", bad_o)
