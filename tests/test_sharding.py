"""Multi-GPU sharding.

CPU part: the deterministic partition every rank computes for itself (host-only contexts), and a
world_size-2 gloo run over that partition.
GPU part, all through the library's own collective path (ShardAllReduce inside the cxk_* entry
points; the transport is a caller-supplied all-reduce here because one box has one GPU, RCCL in
production):
  * G ranks as G threads on one device, reductions through a barrier -- a whole Newton iteration
    (KKT solve, solve-only right-hand sides, PrepareStep, eigenvalue query, step scalars, TakeStep)
    and whole CONEX_Maximize runs must reproduce the single-context results;
  * TWO PROCESSES (torch.distributed, gloo) sharing GPU 0: the real exchange buffers travel through
    gloo and the direction matches the single-context one;
  * the RCCL call path itself (librccl.so loaded on demand, ncclAllReduce on the context's stream)
    through a one-rank communicator.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conex_amd import KktContext
from conex_amd import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _shard_contexts(prob, world, device):
    ctxs = []
    for r in range(world):
        k = KktContext(prob["num_vars"], device=device)
        for c, cl in enumerate(prob["cliques"]):
            k.add_lmi(prob["A"][c], prob["C"][c], cl)
        k.set_shard(r, world)
        k.initialize()
        ctxs.append(k)
    return ctxs


@pytest.mark.parametrize("world", [2, 4, 8])
def test_partition_is_consistent_across_ranks(world):
    prob = syn.lmi_problem(K=200, n=3, m=20, branching=8, overlap=5, seed=1)
    ctxs = _shard_contexts(prob, world, device=-1)
    K, N = ctxs[0].K, ctxs[0].N
    own = np.array([[k.owns(i) for i in range(K)] for k in ctxs])
    assert np.all(own.sum(axis=0) == 1), "every constraint is assembled by exactly one rank"
    infos = [k.shard_info() for k in ctxs]
    assert len(set(infos)) == 1, "cut level / exchange size must agree on all ranks"
    cut, nlev, count = infos[0]
    assert 0 < cut <= nlev and count > 0
    valid = np.array([k.valid_variables() for k in ctxs])
    assert np.all(valid.sum(axis=0) >= 1), "every variable is solved for by some rank"
    top = valid.sum(axis=0) == world
    assert top.sum() > 0 and np.all((valid.sum(axis=0) == 1) | top)
    # balance: no rank owns more than 2x the average number of constraints
    assert own.sum(axis=1).max() <= 1.3 * K / world + 8
    # a constraint only touches variables its owner holds
    for r, k in enumerate(ctxs):
        for i in range(K):
            if own[r, i]:
                assert np.all(valid[r, prob["cliques"][i]])


def test_headline_partition_eight_ranks():
    prob = syn.lmi_problem(K=1000, n=2, m=20, seed=2)  # C4 structure (tiny blocks: host-only)
    ctxs = _shard_contexts(prob, 8, device=-1)
    own = np.array([[k.owns(i) for i in range(1000)] for k in ctxs]).sum(axis=1)
    assert own.sum() == 1000 and own.max() <= 1.15 * 125 + 1
    cut, nlev, count = ctxs[0].shard_info()
    assert count * 8 < 64 * 1024, "only the top of the tree is exchanged (a few KB)"


GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from conex_amd import KktContext
from conex_amd import synthetic as syn
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
prob = syn.lmi_problem(K=60, n=3, m=20, branching=4, overlap=5, seed=3)
k = KktContext(prob["num_vars"], device=-1)
for c, cl in enumerate(prob["cliques"]):
    k.add_lmi(prob["A"][c], prob["C"][c], cl)
k.set_shard(rank, world)
k.initialize()
cut, nlev, count = k.shard_info()
own = torch.tensor([int(k.owns(i)) for i in range(k.K)])
dist.all_reduce(own)
assert bool((own == 1).all())
x = torch.full((count,), float(rank + 1), dtype=torch.float64)   # stand-in exchange buffer
dist.all_reduce(x)
assert bool((x == world * (world + 1) / 2).all())
sizes = [None] * world
dist.all_gather_object(sizes, (cut, nlev, count))
assert len(set(sizes)) == 1
if rank == 0:
    print("GLOO_OK", count)
dist.destroy_process_group()
"""


def test_world_size_2_gloo_exchange(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(GLOO_WORKER)
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    port = 29500 + (os.getpid() % 1000)
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT],
        capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "GLOO_OK" in out.stdout


class ThreadRanks:
    """G ranks as G threads of one process: the all-reduce every rank calls deposits its array,
    waits for the others and reduces the deposits in rank order (deterministic)."""

    def __init__(self, world):
        import threading
        self.world = world
        self.slots = [None] * world
        self.barrier = threading.Barrier(world)
        self.errors = []

    def allreduce(self, rank):
        def fn(arr, op):
            self.slots[rank] = arr.copy()
            self.barrier.wait(timeout=120)
            stack = np.stack(self.slots)
            out = stack.sum(axis=0) if op == 0 else (stack.max(axis=0) if op == 1 else stack.min(axis=0))
            self.barrier.wait(timeout=120)
            return out
        return fn

    def run(self, body):
        import threading
        results = [None] * self.world

        def work(r):
            try:
                results[r] = body(r, self.allreduce(r))
            except BaseException as e:  # noqa: BLE001 -- reported by the main thread
                self.errors.append((r, repr(e)))
                self.barrier.abort()
        threads = [threading.Thread(target=work, args=(r,)) for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=600)
        assert not self.errors, self.errors
        return results


def _program(kind, seed):
    if kind == "lmi":
        prob = syn.lmi_problem(K=120, n=6, m=20, branching=8, overlap=5, seed=seed)
        W = syn.scaling_points(120, 6, seed=4)
    elif kind == "c4":
        prob = syn.lmi_problem(K=300, n=20, m=20, branching=8, overlap=5, seed=seed)   # MFMA Schur kernel, chain at the top
        W = syn.scaling_points(300, 20, seed=4)
    elif kind == "mixed":
        prob = syn.mixed_problem(K=230, seed=seed)
        W = syn.mixed_scaling_points(prob, seed=32)
    elif kind == "chain":
        # config 3's arrangement: second-order cones in a chain, factored in its segment-parallel order
        # (symbolic.h): the pieces are the subtrees the ranks share out
        prob = syn.soc_problem(K=600, dim=10, m=10, overlap=2, seed=seed)
        W = syn.soc_scaling_points(600, 10, seed=33)
    else:
        raise ValueError(kind)
    return prob, W


def _newton_iteration(k, prob, W, owned_only):
    """One full Newton iteration through the ordinary entry points; returns everything comparable."""
    for i in range(k.K):
        if not owned_only or k.owns(i):
            k.set_W(i, W[i])
    out = {}
    ok, out["y"] = k.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    assert ok == 1
    out["eig"] = k.weighted_slack_eigenvalues(None, 0.7 * 0.8)
    out["info"] = k.prepare_step(None, 0.7 * 0.8, 1.0)
    out["scal"] = k.step_scalars()
    k.take_step(min(1.0, 2.0 / out["info"][1] ** 2))
    # a second, separately assembled factorization and two solve-only right-hand sides
    k.assemble()
    assert k.factor() == 1
    k.newton_direction(0.9, 0.8, 0.7)
    out["y2"] = k.get_y()
    rhs = np.random.default_rng(5).uniform(-1, 1, k.N)
    out["y3"] = k.solve_inplace(rhs)
    out["W"] = {i: k.get_W(i) for i in range(k.K) if not owned_only or k.owns(i)}
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("kind,world", [("lmi", 2), ("lmi", 3), ("c4", 4), ("c4", 8), ("mixed", 2), ("mixed", 5), ("chain", 4)])
def test_sharded_newton_iteration_through_the_library_collectives(monkeypatch, kind, world, fused):
    """fused: the factor-and-solve of a rank runs on the whole-tree kernels -- own subtrees up with the
    pack of the exchange buffer behind them, all-reduce, top straight from the buffer and the way back
    down: two launches (tree_fused.h, kFusedShardUp / kFusedShardTop); CXK_NO_FUSED_SHARD=1 keeps the
    level kernels with the separate pack / unpack launches.  Both must reproduce the single context."""
    if fused:
        monkeypatch.delenv("CXK_NO_FUSED_SHARD", raising=False)
    else:
        monkeypatch.setenv("CXK_NO_FUSED_SHARD", "1")
    prob, W = _program(kind, 11)
    build_kind = {"mixed": "mixed", "chain": "soc"}.get(kind, "lmi")
    ref = _newton_iteration(syn.build(KktContext, prob, build_kind, device=0), prob, W, False)

    def body(rank, allreduce):
        k = KktContext(prob["num_vars"], device=0)
        for c, cl in enumerate(prob["cliques"]):
            if build_kind == "lmi":
                k.add_lmi(prob["A"][c], prob["C"][c], cl)
            elif build_kind == "soc":
                k.add_soc(prob["A"][c], prob["c"][c], cl)
            elif prob["kinds"][c] == "herm":
                k.add_hermitian(prob["A"][c], prob["C"][c], cl)
            else:
                k.add_soc(prob["A"][c], prob["C"][c], cl)
        k.set_shard(rank, world)
        k.initialize()
        k.comm_set_allreduce(allreduce)
        k.set_cost(prob["b"])
        out = _newton_iteration(k, prob, W, True)
        out["fused_tree"] = k.fused_tree()
        return out

    for out in ThreadRanks(world).run(body):
        assert out["fused_tree"] == fused
        for key in ("y", "y2", "y3"):
            assert np.linalg.norm(out[key] - ref[key]) <= 1e-10 * np.linalg.norm(ref[key]), key
        assert np.allclose(out["eig"], ref["eig"], rtol=1e-9, atol=0)
        assert np.allclose(out["info"], ref["info"], rtol=1e-9, atol=0)
        assert np.allclose(out["scal"], ref["scal"], rtol=1e-10, atol=1e-12)
        for i, w in out["W"].items():
            assert np.linalg.norm(w - ref["W"][i]) <= 1e-10 * np.linalg.norm(ref["W"][i])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_a_failed_leaf_pivot_on_one_rank_is_seen_by_every_rank(world):
    """A leaf whose scaling point is singular makes its Schur block singular; its supernode is
    factored by ONE rank, inside the first level with the assembly folded in, which reports through
    the tagged flag word.  The flag must cross the exchange: ranks that disagree about Factor()
    take different branches of the IPM loop and issue mismatched collectives."""
    prob, W = _program("c4", 13)
    K = len(prob["cliques"])
    bad = K - 1                                   # a leaf of the clique tree
    Wbad = np.zeros((20, 20))                     # G = 0: the leaf's first pivot is not positive

    def body(rank, allreduce):
        k = KktContext(prob["num_vars"], device=0)
        for c, cl in enumerate(prob["cliques"]):
            k.add_lmi(prob["A"][c], prob["C"][c], cl)
        k.set_shard(rank, world)
        k.initialize()
        k.comm_set_allreduce(allreduce)
        k.set_cost(prob["b"])
        for i in range(k.K):
            if k.owns(i):
                k.set_W(i, W[i])
        ok_good, _ = k.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
        if k.owns(bad):
            k.set_W(bad, Wbad)
        ok_bad, _ = k.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
        if k.owns(bad):
            k.set_W(bad, W[bad])
        ok_again, _ = k.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
        return ok_good, ok_bad, ok_again, bool(k.owns(bad)), k.fused_assembly()

    res = ThreadRanks(world).run(body)
    assert sum(r[3] for r in res) == 1
    for ok_good, ok_bad, ok_again, _, _ in res:
        assert ok_good == 1 and ok_bad == 0 and ok_again == 1


@pytest.mark.gpu
def test_sharded_equality_constraints_take_the_ldlt_path():
    """Multipliers make the KKT matrix indefinite: block LDLT under sharding (the LQR program of
    assembly_test.cc:67-106 with 40 stages) against the single-context solve."""
    from test_oracle_kat import build_lqr_problem
    ref = build_lqr_problem(KktContext, 40, device=0)
    ref.assemble()
    assert ref.factor() == 1
    rhs = np.random.default_rng(2).uniform(-1, 1, ref.N)
    y_ref = ref.solve_inplace(rhs)
    world = 3

    def body(rank, allreduce):
        k = build_lqr_problem(lambda nv, **kw: _sharded(nv, rank, world), 40)
        k.comm_set_allreduce(allreduce)
        k.assemble()
        assert k.factor() == 1
        return k.solve_inplace(rhs)

    def _sharded(nv, rank, world):
        k = KktContext(nv, device=0)
        k.set_shard(rank, world)
        return k

    for y in ThreadRanks(world).run(body):
        assert np.linalg.norm(y - y_ref) <= 1e-10 * np.linalg.norm(y_ref)


@pytest.mark.gpu
def test_sharded_conex_maximize_matches_single_gpu():
    """CONEX_Maximize with a communicator set runs the sharded IPM loop: every rank builds the same
    program, returns the same y, and that y is the single-GPU optimum."""
    import ctypes as C
    import conex_api as ca
    prob = syn.lmi_problem(K=100, n=20, m=20, branching=8, overlap=5, seed=21)
    L = ca.api()
    L.CONEX_HIP_SetAllReduce.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    fn_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p)
    hip = C.CDLL("libamdhip64.so")

    def solve(rank, world, allreduce):
        p = L.CONEX_CreateConeProgram()
        keep = None
        if world > 1:
            def cb(user, dev, count, op, stream):
                buf = np.empty(count)
                hip.hipStreamSynchronize(C.c_void_p(stream))
                hip.hipMemcpy(buf.ctypes.data_as(C.c_void_p), C.c_void_p(dev), C.c_size_t(8 * count), 2)
                out = np.ascontiguousarray(allreduce(buf, op))
                hip.hipMemcpy(C.c_void_p(dev), out.ctypes.data_as(C.c_void_p), C.c_size_t(8 * count), 1)
                return 0
            keep = fn_t(cb)
            assert L.CONEX_HIP_SetAllReduce(p, rank, world, C.cast(keep, C.c_void_p), None) == 0
        assert L.CONEX_SetNumberOfVariables(p, prob["num_vars"]) == 0
        for c, cl in enumerate(prob["cliques"]):
            a, cm = ca.colmajor(prob["A"][c]), ca.colmajor(prob["C"][c])
            v = np.ascontiguousarray(cl, dtype=np.int64)
            assert L.CONEX_AddSparseLMIConstraint(p, ca.dp(a), 20, 20, 20, ca.dp(cm), 20, 20,
                                                  v.ctypes.data_as(C.POINTER(C.c_long)), 20) == c
        cfg = ca.default_config()
        y = np.zeros(prob["num_vars"])
        b = np.ascontiguousarray(prob["b"])
        ok = L.CONEX_Maximize(p, ca.dp(b), len(b), C.byref(cfg), ca.dp(y), len(y))
        st = ca.IterationStats()
        L.CONEX_GetIterationStats(p, C.byref(st), -1)
        L.CONEX_DeleteConeProgram(p)
        return ok, y, st.iteration_number + 1

    ok0, y0, it0 = solve(0, 1, None)
    assert ok0 == 1
    world = 4
    results = ThreadRanks(world).run(lambda r, ar: solve(r, world, ar))
    for ok, y, it in results:
        assert ok == 1 and abs(it - it0) <= 3
        assert np.array_equal(y, results[0][1])           # every rank returns the same vector
        assert abs(prob["b"] @ y - prob["b"] @ y0) <= 1e-6 * abs(prob["b"] @ y0)


@pytest.mark.gpu
def test_rccl_call_path_with_a_one_rank_communicator():
    """librccl.so loaded on demand, ncclCommInitRank, ncclAllReduce (sum, max, min) of device
    memory on the context's stream.  (Two ranks cannot share the one GPU of this box; the driver's
    multi-GPU bench runs the same code with world > 1.)"""
    prob = syn.lmi_problem(K=5, n=4, m=4, branching=2, overlap=2, seed=3)
    k = KktContext(prob["num_vars"], device=0)
    for c, cl in enumerate(prob["cliques"]):
        k.add_lmi(prob["A"][c], prob["C"][c], cl)
    k.comm_init_rccl(KktContext.comm_unique_id(), 0, 1)
    k.initialize()
    k.comm_selftest(5000)
    ok, y = k.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    assert ok == 1 and np.all(np.isfinite(y))


TWO_PROCESS_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from conex_amd import KktContext
from conex_amd import synthetic as syn
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
prob = syn.lmi_problem(K=120, n=20, m=20, branching=8, overlap=5, seed=3)
W = syn.scaling_points(120, 20, seed=4)
k = KktContext(prob["num_vars"], device=0)            # both processes share GPU 0
for c, cl in enumerate(prob["cliques"]):
    k.add_lmi(prob["A"][c], prob["C"][c], cl)
k.set_shard(rank, world)
k.initialize()
def allreduce(arr, op):                                 # the REAL exchange buffers travel through gloo
    t = torch.from_numpy(arr.copy())
    dist.all_reduce(t, op={0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}[op])
    return t.numpy()
k.comm_set_allreduce(allreduce)
for i in range(k.K):
    if k.owns(i):
        k.set_W(i, W[i])
ok, y = k.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
assert ok == 1
info = k.prepare_step(None, 0.56, 1.0)
if rank == 0:
    ref = syn.build(KktContext, prob, "lmi", device=0)
    for i in range(ref.K):
        ref.set_W(i, W[i])
    okr, yr = ref.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    ir = ref.prepare_step(None, 0.56, 1.0)
    err = np.linalg.norm(y - yr) / np.linalg.norm(yr)
    assert okr == 1 and err <= 1e-10, err
    assert np.allclose(info, ir, rtol=1e-9, atol=0)
    print("TWO_PROCESS_OK", err)
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_two_processes_exchange_the_real_buffers_through_gloo(tmp_path):
    script = tmp_path / "worker2.py"
    script.write_text(TWO_PROCESS_WORKER)
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    port = 29600 + (os.getpid() % 1000)
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT],
        capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "TWO_PROCESS_OK" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("world,K,sparse", [(2, 40, False), (3, 100, False), (8, 300, False), (4, 120, True)])
def test_sharded_direction_equals_single_gpu(monkeypatch, world, K, sparse):
    prob = syn.lmi_problem(K=K, n=6, m=20, branching=8, overlap=5, seed=11)
    if sparse:  # the shards assemble their constraints through the sparse-LMI kernels
        prob = syn.sparsify(prob, 0.15)
        monkeypatch.setenv("CXK_SPARSE_LMI", "1")
    W = syn.scaling_points(K, 6, seed=4)
    ref = syn.build(KktContext, prob, "lmi", device=0)
    for i in range(K):
        ref.set_W(i, W[i])
    ok, y_ref = ref.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
    assert ok == 1
    ctxs = _shard_contexts(prob, world, device=0)
    for k in ctxs:
        for i in range(K):
            if k.owns(i):
                k.set_W(i, W[i])
        k.set_cost(prob["b"])
        k.kkt_local_async(0.7, 0.9, 0.8)
    for k in ctxs:
        assert k.sync() == 1
    total = sum(k.exchange_download() for k in ctxs)   # what all_reduce(SUM) delivers to every rank
    for k in ctxs:
        k.exchange_upload(total)
    for k in ctxs:
        k.kkt_finish_async(0.7, 0.9, 0.8)
    for k in ctxs:
        assert k.sync() == 1
        y = k.get_y()
        v = k.valid_variables()
        err = np.linalg.norm(y[v] - y_ref[v]) / np.linalg.norm(y_ref[v])
        assert err <= 1e-10, err
