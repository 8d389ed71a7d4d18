"""Wall time per call of the KKT solve, the solve-only path and the cone-update queries on the
C4 program (run on the GPU box): python tools/solve_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from conex_amd import KktContext, synthetic as syn
prob = syn.lmi_problem(K=1000, n=20, m=20, branching=8, overlap=5)
W = syn.scaling_points(1000, 20)
ctx = KktContext(prob["num_vars"], device=0)
for c, cl in enumerate(prob["cliques"]):
    ctx.add_lmi(prob["A"][c], prob["C"][c], cl)
ctx.initialize()
for i in range(ctx.K):
    ctx.set_W(i, W[i])
ctx.set_cost(prob["b"])
ctx.kkt_solve_async(0.7, 0.9, 0.8); ctx.sync()
def timeit(fn, n=200):
    for _ in range(20): fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    ctx.sync()
    return (time.perf_counter() - t0) / n * 1e6
print("kkt_solve_async   %.1f us" % timeit(lambda: ctx.kkt_solve_async(0.7, 0.9, 0.8)))
print("solve_rhs         %.1f us" % timeit(lambda: ctx.solve_rhs(-0.9, 0.8, 0.0)))
print("prepare_step      %.1f us" % timeit(lambda: ctx.prepare_step(None, 0.56, 1.0)))
print("eigenvalues       %.1f us" % timeit(lambda: ctx.weighted_slack_eigenvalues(None, 0.56)))
