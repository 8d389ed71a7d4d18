"""Multi-GPU sharding.

CPU part: the deterministic partition every rank computes for itself (host-only contexts), and
a world_size-2 gloo exchange of a buffer of the real exchange size.
GPU part: G ranks emulated on one device -- local phases, a manual sum of the exchange buffers
(what RCCL all-reduce does), finish phases -- must reproduce the single-context Newton direction.
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
