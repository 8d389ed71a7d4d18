#!/usr/bin/env python3
"""Headline benchmark: Newton KKT-solves/sec on BASELINE.json config 4.

One "step" = one KKT solve of the chordal SDP with 1000 dense LMIs of order 20
(N = 15005): dense-LMI Schur assembly -> gather into the supernodal slab -> supernodal
Cholesky -> right-hand side -> forward/backward block solves, all on device-resident data
(reference cone_program.cc:338-341, 360, 409-413).

    python bench.py --gpus N --steps K --warmup W

Protocol ("median and min reported"): W untimed warmup steps, then the K-step timed
region -- barrier + synchronize on both sides, max over ranks -- is REPEATED (at least --repeats
times and until --min-timed-ms of timed work has run); `value` = K / median region time, the
regions' min / max / count ride along in `regions`.  No untimed pre-heating: a region that caught
the clock ramp of a freshly started process shows up as `regions.max`.

N > 1: one process per GPU.  Started under torch.distributed.run the script is one rank; started
plainly (`python bench.py --gpus N`) it launches torch.distributed.run on itself as a child
process BEFORE anything touches a GPU, relays the ranks' output and exits with their status.
The constraints and elimination subtrees of the ONE program are sharded across ranks (strong
scaling); the only thing that touches the GPUs' links is the library's own RCCL
communicator (a `gloo` group on the CPU ships its 128-byte id and carries the timing barrier).
`value` is ALWAYS the sharded layout's rate; the same step unsharded on every rank is timed beside
it and reported as `replicated_value` (one rank's rate, not multiplied by N).  Prints one JSON line
on rank 0.
"""
import argparse
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_MFMA_PEAK_TF = 78.6  # dense fp64 matrix peak


def git_blob_hash(path):
    """`git hash-object` of a file: names the exact source a committed measurement belongs to."""
    try:
        data = open(path, "rb").read()
    except OSError:
        return None
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


SCHUR_SOURCE = os.path.join(ROOT, "conex_amd", "csrc", "lmi_fused_mfma.hip")


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE x2 correction + WRITE_SIZE, profiles/rNN/pmc_summary.json) together with the git
    blob hash of the kernel source it was measured on; (None, None, None) if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_summary.json")))
    if not files:
        return None, None, None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        for key in ("lmi_schur_mfma", "lmi_schur_fused"):
            if key in d:
                return (float(d[key]["hbm_traffic_bytes_per_launch"]), d[key].get("kernel_source_blob"),
                        os.path.relpath(files[-1], ROOT))
    except Exception:
        pass
    return None, None, None


def cpu_baseline(prob, W, budget_s=12.0, kind="lmi"):
    """The oracle (plain-C port of the reference path, 1 thread) on the same workload."""
    import oracle_lib as ol
    from conex_amd import synthetic as syn
    o = syn.build(ol.Program, prob, kind)
    for i in range(o.K):
        o.set_W(i, W[i])
    t0 = time.perf_counter()
    n = 0
    while True:
        ok, y = o.kkt_solve(prob["b"], 0.7, 0.9, 0.8)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 200:
            break
    return {"value": n / el, "unit": "KKT-solves/s", "cores": 1, "kind": "port",
            "host_cores": os.cpu_count(),
            "sample": f"{n} full KKT-solves of the same workload ({o.K} constraints), 1 thread of "
                      f"{os.cpu_count()} host cores, {el:.1f} s"}, y


def newton_step_section(prob, device, ctx_stats):
    """SURVEY 8(d)'s secondary metric: the FULL Newton step of the interior-point loop -- assemble,
    factor, mu selection (eigenvalue query + one solve), Newton direction, PrepareStep, TakeStep
    (cone_program.cc:338-436) -- as CONEX_Maximize runs it through conex.h on the same program:
    wall time per iteration of whole solves, and the device time of each kernel of an iteration from
    hipEvent pairs (cxk_kernel_clock) with the algorithmic bytes it is priced against."""
    import ctypes as C
    import numpy as np
    from conex_amd import capi as ca
    L = ca.api()
    L.CONEX_HIP_SetDevice.argtypes = [C.c_void_p, C.c_int]
    L.CONEX_HIP_KernelClocks.argtypes = [C.c_void_p, C.c_int]
    L.CONEX_HIP_ReadKernelClocks.argtypes = [C.c_void_p, ca.c_double_p, C.POINTER(C.c_int)]
    n, m = prob["n"], prob["m"]
    K = len(prob["cliques"])
    p = L.CONEX_CreateConeProgram()
    L.CONEX_HIP_SetDevice(p, device)
    assert L.CONEX_SetNumberOfVariables(p, prob["num_vars"]) == 0
    for c, cl in enumerate(prob["cliques"]):
        a, cm = ca.colmajor(prob["A"][c]), ca.colmajor(prob["C"][c])
        v = np.ascontiguousarray(cl, dtype=np.int64)
        assert L.CONEX_AddSparseLMIConstraint(p, ca.dp(a), n, n, len(cl), ca.dp(cm), n, n,
                                              v.ctypes.data_as(C.POINTER(C.c_long)), len(cl)) == c
    b = np.ascontiguousarray(prob["b"], dtype=np.float64)
    y = np.zeros(len(b))
    walls, iters_seen = [], []
    st = ca.IterationStats()

    def solve():
        cfg = ca.default_config()
        t0 = time.perf_counter()
        ok = L.CONEX_Maximize(p, ca.dp(b), len(b), C.byref(cfg), ca.dp(y), len(b))
        dt = time.perf_counter() - t0
        L.CONEX_GetIterationStats(p, C.byref(st), -1)
        return ok, st.iteration_number + 1, dt

    ok0, _, _ = solve()   # set-up (symbolic analysis, upload) + a first solve: not timed
    for _ in range(5):
        ok, it, dt = solve()
        walls.append(1e6 * dt / max(it, 1))
        iters_seen.append(it)
        ok0 = ok0 and ok
    L.CONEX_HIP_KernelClocks(p, 1)
    ok, it, dt_clocked = solve()
    avg = np.zeros(6)
    cnt = np.zeros(6, dtype=np.int32)
    L.CONEX_HIP_ReadKernelClocks(p, ca.dp(avg), cnt.ctypes.data_as(C.POINTER(C.c_int)))
    L.CONEX_HIP_KernelClocks(p, 0)
    L.CONEX_DeleteConeProgram(p)
    tri = m * n * (n + 1) // 2           # the A_i as packed lower triangles (what the step kernels stream)
    slab_b, N = ctx_stats["slab_bytes"], ctx_stats["N"]
    work = {   # algorithmic bytes per launch (SURVEY 8d formulas)
        "assembly": ("lmi_schur_mfma", 8.0 * K * (m * n * n + 2 * n * n + m * (m + 1) // 2 + 2 * m)),
        "tree": ("tree_fused (gather + factor + first solve)", 8.0 * K * m * m + 2 * slab_b + slab_b + 32.0 * N),
        "query": ("lmi_prepare_rows<1> (eigenvalue query)", 8.0 * K * (tri + 2 * n * n)),
        "solve": ("tree_fused_solve (solve-only sweep)", slab_b + 32.0 * N),
        "prepare": ("lmi_prepare_rows<0> (PrepareStep)", 8.0 * K * (tri + 3 * n * n)),
        "take": ("lmi_take_step_rows (TakeStep)", 8.0 * K * 3 * n * n),
    }
    order = ["assembly", "tree", "solve", "query", "prepare", "take"]   # CXK_CLOCK_* order
    kernels = {}
    total_us = 0.0
    for slot, key in enumerate(order):
        name, nbytes = work[key]
        us = 1e3 * float(avg[slot])
        if cnt[slot] == 0:
            continue
        per_iter = cnt[slot] / max(it, 1)
        total_us += us * per_iter
        kernels[name] = {"us": us, "launches_per_iteration": per_iter, "algorithmic_bytes": nbytes,
                         "achieved_GBps": nbytes / (us * 1e-6) / 1e9 if us > 0 else None,
                         "frac": nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS if us > 0 else None, "bound": "hbm"}
    return {"us_per_iteration": statistics.median(walls), "us_per_iteration_min": min(walls),
            "us_per_iteration_max": max(walls), "solves_timed": len(walls), "iterations_per_solve": iters_seen,
            "solved": bool(ok0), "kernel_us_per_iteration": total_us, "kernels": kernels,
            "unit": "us", "through": "CONEX_Maximize (include/conex.h), whole solves from W = I to mu = 1e-6",
            "note": "wall time of a solve / its iterations (set-up excluded: the program's device context "
                    "exists); kernel times from hipEvent pairs on the dispatch in one extra, instrumented solve"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=7,
                    help="the K-step timed region is repeated at least this many times; value = K / median")
    ap.add_argument("--min-timed-ms", type=float, default=30.0,
                    help="... and until this much timed work has run (a 20-step region of config 4 is 1 ms: seven "
                         "of them would be over before a freshly started process has its clocks up)")
    ap.add_argument("--preheat", type=int, default=0,
                    help="extra untimed steps before the --warmup steps (default 0; reported as config.preheat_steps)")
    ap.add_argument("--K", type=int, default=1000)
    ap.add_argument("--workload", choices=["c4", "c2", "c3", "c5", "c4s", "maxcut"], default="c4",
                    help="c4 (default, the metric's config): 1000 LMIs n=20; c5 mixed complex Hermitian "
                         "+ SOC tree, N = 50k (the config BASELINE names for 8-way sharding); extras, single "
                         "GPU: c2 one LMI n=200 m=50 (MFMA-bound assembly); c3 5000 second-order cones in a "
                         "chain (no tree parallelism: latency-bound sweeps); c4s the C4 structure with sparse "
                         "A_i (--density), the sparse-LMI evaluation path; maxcut one LMI of order --maxcut-n "
                         "with one-nonzero A_i (sparse assembly + one big supernode)")
    ap.add_argument("--maxcut-n", type=int, default=200)
    ap.add_argument("--density", type=float, default=0.005,
                    help="with --workload c4s: fraction of the entries of every A_i that is nonzero")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-newton-step", action="store_true",
                    help="skip the secondary metric (full interior-point iterations through conex.h)")
    ap.add_argument("--shard-path", action="store_true",
                    help="run the sharded step (own subtrees / all-reduce / top and back) even at "
                         "world size 1: exercises the multi-GPU code path on a single-GPU box")
    ap.add_argument("--shard-world", type=int, default=2,
                    help="with --shard-path: size of the virtual world this GPU is rank 0 of")
    ap.add_argument("--soc-tree", type=int, default=0,
                    help="with --workload c3: arrange the cones in a b-ary clique tree instead of a chain")
    ap.add_argument("--cold-copies", type=int, default=5,
                    help="c4, one GPU: also time the step cycling this many copies of the program "
                         "(5 x 64 MB of A > the 256 MiB Infinity Cache), reported as \"cold_cache\"; 0/1 = off")
    ap.add_argument("--event-samples", type=int, default=20,
                    help="launches of the step's kernels timed with a hipEvent pair in a separate pass "
                         "AFTER the timed regions (at most --steps)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain start: become the launcher.  Nothing in this process has touched a GPU (no HIP call,
        # not even `import torch`), the ranks are children, their status is ours.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL needs dmabuf IPC on this host driver
        raise SystemExit(subprocess.call(cmd, env=env))

    # stdout carries ONE JSON line and nothing else: libraries that print banners there (gloo, RCCL's
    # version block at communicator creation) get stderr instead for the life of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    from conex_amd import KktContext
    from conex_amd import synthetic as syn

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE)")
    ndev = torch.cuda.device_count()   # (counting devices does not initialise the GPU)
    if ndev < max(world, 1) or local_rank >= ndev:
        raise SystemExit(f"rank {rank}: need {world} devices for --gpus {world}, this node shows {ndev}")
    torch.cuda.set_device(local_rank)
    dist = None
    sharded = world > 1 or args.shard_path
    if sharded:
        # control plane only: the unique id of the library's RCCL communicator and the barriers /
        # max over ranks of the timing travel over gloo on the CPU; no torch process group on the GPUs
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    kind = "lmi"
    if args.workload not in ("c4", "c5", "c3") and world > 1:
        raise SystemExit(f"--workload {args.workload} is a single-GPU measurement (c4, c5 and c3 shard)")
    if args.workload == "c2":
        args.K, n_order, m_vars = 1, 200, 50
        prob = syn.lmi_problem(K=1, n=200, m=50, branching=2, overlap=1)
        W = syn.scaling_points(args.K, n_order)
    elif args.workload == "c3":
        kind, args.K, n_order, m_vars = "soc", 5000, 10, 10
        prob = syn.soc_problem(K=5000, dim=10, m=10, overlap=2, tree=args.soc_tree)
        W = syn.soc_scaling_points(5000, 10)
    elif args.workload == "maxcut":
        args.K, n_order, m_vars = 1, args.maxcut_n, args.maxcut_n
        prob = syn.maxcut_problem(args.maxcut_n)
        W = syn.scaling_points(1, n_order, scale=0.05)
    elif args.workload == "c5":
        kind, n_order, m_vars = "mixed", 12, 24
        prob = syn.mixed_problem()
        args.K = len(prob["cliques"])
        W = syn.mixed_scaling_points(prob)
    else:
        n_order, m_vars = 20, 20
        prob = syn.lmi_problem(K=args.K, n=20, m=20, branching=8, overlap=5)
        if args.workload == "c4s":
            prob = syn.sparsify(prob, args.density)
        W = syn.scaling_points(args.K, n_order)
    stream = torch.cuda.current_stream().cuda_stream

    def torch_allreduce(arr, op):
        # fallback transport (see below): host copy -> device tensor -> torch.distributed -> back
        t = torch.from_numpy(np.ascontiguousarray(arr))
        dist.all_reduce(t, op=(dist.ReduceOp.SUM, dist.ReduceOp.MAX, dist.ReduceOp.MIN)[op])
        return t.numpy()

    def build_context(collective):
        ctx = KktContext(prob["num_vars"], device=local_rank, stream=stream)
        for c, cl in enumerate(prob["cliques"]):
            if kind == "lmi":
                ctx.add_lmi(prob["A"][c], prob["C"][c], cl)
            elif kind == "soc":
                ctx.add_soc(prob["A"][c], prob["c"][c], cl)
            elif prob["kinds"][c] == "herm":
                ctx.add_hermitian(prob["A"][c], prob["C"][c], cl)
            else:
                ctx.add_soc(prob["A"][c], prob["C"][c], cl)
        if collective == "rccl":
            # the library's own RCCL communicator: rank 0 makes the unique id, torch.distributed only
            # carries the 128 bytes (and the timing barrier below); every collective of the step --
            # the all-reduce of the packed top of the tree -- is issued by libconex.so on its stream
            uid = [KktContext.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            ctx.comm_init_rccl(uid[0], rank, world)
        elif collective == "torch":
            ctx.set_shard(rank, world)
        elif collective == "none":
            # --shard-path on one GPU: rank 0 of a virtual --shard-world-rank world whose all-reduces
            # run on a ONE-rank RCCL communicator (the other shards are missing from the exchange,
            # so only timing and the plumbing are meaningful, not the direction)
            ctx.set_shard(0, args.shard_world)
        ctx.initialize()
        if collective == "torch":
            ctx.comm_set_allreduce(torch_allreduce)
        elif collective == "none":
            # real ncclAllReduce calls on a one-rank communicator (they return their input)
            ctx.comm_init_rccl_solo()
        return ctx

    def load_state(c, all_constraints):
        for i in range(c.K):
            if all_constraints or c.owns(i):
                c.set_W(i, W[i])
        c.set_cost(prob["b"])

    collective = "rccl" if world > 1 else ("none" if sharded else "")
    try:
        ctx = build_context(collective)
    except Exception as e:  # e.g. librccl.so not loadable from the library: same failure on every rank
        if collective != "rccl":
            raise
        print("rank %d: in-library RCCL communicator failed (%s); falling back to torch.distributed "
              "through the all-reduce callback" % (rank, e), file=sys.stderr, flush=True)
        collective = "torch"
        ctx = build_context(collective)
    load_state(ctx, not sharded)
    exch_bytes = 8 * ctx.shard_info()[2] if sharded else 0

    def fence():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_regions(c, steps, repeats, min_total_s, cap=400):
        """K-step regions, each bracketed by barrier + synchronize on both sides, max over ranks."""
        times = []
        okc = True
        while len(times) < repeats or (sum(times) < min_total_s and len(times) < cap):
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                c.kkt_solve_async(0.7, 0.9, 0.8)
            fence()
            el = time.perf_counter() - t0
            okc = c.sync() and okc
            if sharded:
                t = torch.tensor([el], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)     # every rank sees the same number: same loop count
                el = float(t.item())
            times.append(el)
        return times, okc

    for _ in range(max(0, args.preheat) + args.warmup):
        ctx.kkt_solve_async(0.7, 0.9, 0.8)
    ok = ctx.sync()
    times, okr = timed_regions(ctx, args.steps, args.repeats, 1e-3 * args.min_timed_ms)
    ok = ok and okr
    med = statistics.median(times)

    # N > 1 (and the one-GPU rehearsal): the SAME step unsharded on every rank, timed beside the sharded
    # one.  `value` stays the sharded layout's; the replicated rate is one rank's rate (NOT multiplied
    # by N): what the job would get by ignoring the other GPUs.
    replicated = None
    if sharded:
        rep = build_context("")
        load_state(rep, True)
        for _ in range(args.warmup):
            rep.kkt_solve_async(0.7, 0.9, 0.8)
        rep.sync()
        rt, rok = timed_regions(rep, args.steps, min(args.repeats, 5), 0.5e-3 * args.min_timed_ms)
        rmed = statistics.median(rt)
        replicated = {"value": args.steps / rmed, "us_per_step": 1e6 * rmed / args.steps,
                      "us_per_step_min": 1e6 * min(rt) / args.steps, "regions": len(rt), "factor_ok": bool(rok)}
        del rep

    # the kernels' durations: a SEPARATE, untimed pass of the same step with a hipEvent pair on every
    # launch of the assembly kernel and of the tree launch (a bracketed launch costs the stream a
    # bubble, which is why it stays out of the timed regions)
    ctx.enable_timing(1)
    for name in ("assembly", "tree"):
        ctx.kernel_clock(name, reset=True)
    for _ in range(max(1, min(args.event_samples, args.steps))):
        ctx.kkt_solve_async(0.7, 0.9, 0.8)
    ok = ctx.sync() and ok
    ctx.enable_timing(False)
    nsamp, kern_ms = ctx.kernel_clock("assembly", reset=True)
    ntree, tree_ms = ctx.kernel_clock("tree", reset=True)
    abytes, aflops = ctx.assembly_work()
    y = ctx.get_y() if not sharded else None
    slab_bytes = 8.0 * ctx.slab_size()

    if rank == 0:
        out = {
            "metric": {"c4": "Newton KKT-solves/sec (assemble+factor+solve), 1000x(20x20) PSD blocks, fp64",
                       "c4s": "Newton KKT-solves/sec (assemble+factor+solve), 1000x(20x20) PSD blocks with sparse A_i, fp64",
                       "maxcut": f"Newton KKT-solves/sec (assemble+factor+solve), max-cut SDP n=m={args.maxcut_n}, fp64",
                       "c2": "Newton KKT-solves/sec (assemble+factor+solve), one 200x200 PSD block m=50, fp64",
                       "c3": "Newton KKT-solves/sec (assemble+factor+solve), 5000 second-order cones dim 10 ("
                             + (f"{args.soc_tree}-ary tree" if args.soc_tree else "chain") + "), fp64",
                       "c5": "Newton KKT-solves/sec (assemble+factor+solve), 1600 complex 12x12 PSD + 3000 SOC, fp64",
                       }[args.workload],
            "value": args.steps / med,
            "unit": "KKT-solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * med / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "regions": {"count": len(times), "steps_each": args.steps,
                        "median_ms": 1e3 * med, "min_ms": 1e3 * min(times), "max_ms": 1e3 * max(times),
                        "value_median": args.steps / med, "value_min": args.steps / max(times),
                        "value_max": args.steps / min(times),
                        "spread": (max(times) - min(times)) / med,
                        "spread_without_slowest": ((sorted(times)[-2] - min(times)) / med) if len(times) > 2 else None,
                        "note": "value = steps / median region; every region is exactly `steps` steps between "
                                "barrier + synchronize pairs, max over ranks"},
            "config": {"workload": {"c4": "BASELINE config 4: chordal SDP, 1000 dense LMIs n=20 m=20, "
                                          "8-ary clique tree overlap 5, N=15005",
                                    "maxcut": f"max-cut relaxation: one LMI of order {args.maxcut_n} over {args.maxcut_n} "
                                              f"variables, A_i = -e_i e_i^T ({ctx.count_sparse_lmi()} constraint on the "
                                              "sparse path), one dense supernode",
                                    "c4s": f"config 4 structure with sparse A_i (density {args.density}, "
                                           f"{ctx.count_sparse_lmi()} of {args.K} constraints on the sparse path)",
                                    "c2": "BASELINE config 2: one dense LMI n=200, m=50 (profile_sdp.cc shape)",
                                    "c3": "BASELINE config 3: 5000 SOC dim 10, overlap 2, " + (f"{args.soc_tree}-ary clique tree" if args.soc_tree else "chain (N=40002)"),
                                    "c5": "BASELINE config 5: 1600 complex Hermitian PSD order 12 (m=24) + 3000 SOC "
                                          "dim 10, 8-ary clique tree overlap 4, N=50004"}[args.workload],
                       "K": args.K, "n": n_order, "m": m_vars, "N": ctx.N,
                       "parallelism": (f"elimination-subtree sharding x{world if world > 1 else args.shard_world}, one "
                                       f"all-reduce of {exch_bytes} B per solve, "
                                       + ("RCCL, issued by libconex.so" if collective == "rccl" else
                                          "torch.distributed through the all-reduce callback (FALLBACK)"
                                          if collective == "torch" else
                                          f"one-rank RCCL communicator (single-GPU rehearsal as rank 0 of {args.shard_world})"))
                       if sharded else "single GPU",
                       "fused_tree": bool(ctx.fused_tree()),
                       "n_ranks_seen": ctx.comm_count() if sharded else 1,
                       "preheat_steps": max(0, args.preheat),
                       "factor_ok": bool(ok)},
        }
        if sharded:
            out["sharded_value"] = out["value"]
            out["replicated_value"] = replicated["value"]
            out["layouts"] = {"sharded_us_per_step": 1e6 * med / args.steps,
                              "sharded_us_per_step_min": 1e6 * min(times) / args.steps,
                              "replicated_us_per_step": replicated["us_per_step"],
                              "replicated_us_per_step_min": replicated["us_per_step_min"],
                              "exchange_bytes_per_solve": exch_bytes,
                              "note": "`value` = the sharded layout (one program across the ranks); replicated = "
                                      "every rank runs the whole step, the job's rate is one rank's"}
        if nsamp > 0 and kern_ms > 0:
            gbs = abytes / (kern_ms * 1e-3) / 1e9
            if args.workload == "maxcut":
                n_ = float(args.maxcut_n)
                sbytes = 8.0 * (3 * n_ * n_ + n_ * (n_ + 1) / 2)   # W, C, X = W C W, G
                gbs = sbytes / (kern_ms * 1e-3) / 1e9
                out["roofline"] = {"bound": "hbm", "kernel": "sparse LMI assembly (2 GEMM + lmi_schur_sparse)",
                                   "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": gbs / HBM_PEAK_GBS, "traffic": None, "kernel_ms": kern_ms,
                                   "kernel_samples": nsamp, "algorithmic_bytes": sbytes,
                                   "note": "the dense evaluation of the same constraint streams "
                                           f"{8 * n_ ** 3 / 1e6:.0f} MB and does {4 * n_ ** 4 / 1e9:.0f} GFLOP"}
            elif args.workload == "c4s":
                nnz = float(np.count_nonzero(prob["A"]))
                sbytes = 12.0 * nnz + args.K * 8.0 * (2 * 400 + 210 + 42)
                gbs = sbytes / (kern_ms * 1e-3) / 1e9
                out["roofline"] = {"bound": "hbm", "kernel": "lmi_schur_sparse", "achieved": gbs,
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                   "traffic": None, "kernel_ms": kern_ms, "kernel_samples": nsamp,
                                   "algorithmic_bytes": sbytes,
                                   "note": "bytes = nonzeros (12 B each) + W, C, outputs; the kernel is "
                                           "latency-bound at this size, the dense path streams 72.4 MB"}
            elif args.workload == "c3":
                # second-order cones only: no LMI kernel is timed; the step is bound by the depth of
                # the elimination tree (latency), reported as achieved bytes of the whole step
                pass
            elif args.workload == "c5":
                hbytes = abytes / 2.0   # the folded form reads the top half of every real representation
                out["roofline"] = {"bound": "hbm", "kernel": "Hermitian assembly (lmi_schur_mfma<24, H>)",
                                   "achieved": hbytes / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": hbytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                   "kernel_ms": kern_ms, "kernel_samples": nsamp, "algorithmic_bytes": hbytes,
                                   "note": "bytes = the plane-wise data of the reference (real and imaginary "
                                           "parts once): the kernel reads the top half of the order-24 real "
                                           "representation, which is exactly that"
                                           + (" (this rank's share of the constraints)" if sharded else "")}
            elif args.workload == "c2":
                tf = aflops / (kern_ms * 1e-3) / 1e12
                out["roofline"] = {"bound": "mfma", "kernel": "lmi assembly (gemm_f64_mfma x3 + finalize)",
                                   "achieved": tf, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                   "frac": tf / FP64_MFMA_PEAK_TF, "traffic": None,
                                   "kernel_ms": kern_ms, "kernel_samples": nsamp,
                                   "algorithmic_bytes": abytes, "algorithmic_gflop": aflops / 1e9,
                                   "note": "algorithmic flops = the reference's 4n^3(m+1)+n^2(m^2+3m+4) "
                                           "(SURVEY 8d); the kernel executes about half of them "
                                           "(W A_i W is never formed)"}
            else:
                traffic, blob_then, where = pmc_traffic() if (args.K == 1000 and not sharded) else (None, None, None)
                blob_now = git_blob_hash(SCHUR_SOURCE)
                out["roofline"] = {"bound": "hbm", "kernel": "lmi_schur_mfma", "achieved": gbs,
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                   "traffic": traffic,
                                   "traffic_source": {"file": where, "kernel_source_blob_then": blob_then,
                                                      "kernel_source_blob_now": blob_now,
                                                      "stale": (blob_then != blob_now) if traffic is not None else None,
                                                      "what": "rocprofv3 PMC passes committed under profiles/ "
                                                              "(2 x FETCH_SIZE + WRITE_SIZE); the blob hashes are "
                                                              "`git hash-object conex_amd/csrc/lmi_fused_mfma.hip` "
                                                              "when the counters were collected and now"},
                                   "kernel_ms": kern_ms, "kernel_samples": nsamp,
                                   "algorithmic_bytes": abytes, "algorithmic_gflop": aflops / 1e9,
                                   "achieved_tflops": aflops / (kern_ms * 1e-3) / 1e12,
                                   "fp64_mfma_peak_measured_tflops": 77.5,
                                   "note": "the kernel sits at the fp64 ridge: 72.4 MB need 11.5 us at the "
                                           "6.3 TB/s this chip streams (9 us at the 8 TB/s spec priced here), its "
                                           "multiply-adds 10 us of the fp64 pipe (MFMA and VALU share it: "
                                           "profiles/r02/mfma_f64_peak.jsonl)"
                                           + (" (this rank's share of the constraints)" if sharded else "")}
        if ntree > 0 and tree_ms > 0 and args.workload in ("c4", "c5", "c3", "c4s"):
            # the other kernel of the step: assembly gather + factorization + first solve over the whole
            # elimination tree.  Bytes: the Schur arena read once, the slab read and written by
            # the factorization, slab + 4 N doubles by the solve.  It is latency-bound: the critical path is
            # one dependent hand-off + elimination per level up and one hand-off per level down
            # (profiles/r04/fused_tree_stamps.txt), which `critical_path_levels` names.
            g_bytes = 8.0 * sum(len(cl) ** 2 for cl in prob["cliques"])
            tbytes = g_bytes + 3 * slab_bytes + 32.0 * ctx.N
            out["roofline_tree"] = {"bound": "hbm", "kernel": "tree_fused" if ctx.fused_tree() else "tree level kernels",
                                    "achieved": tbytes / (tree_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": tbytes / (tree_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                    "kernel_ms": tree_ms, "kernel_samples": ntree, "algorithmic_bytes": tbytes,
                                    "critical_path_levels": ctx.num_levels(),
                                    "note": "latency-bound by the depth of the elimination tree, not by bytes: per level "
                                            "one cross-CU hand-off (~1.1-1.4 us) + one lone-wavefront elimination "
                                            "(~2 us) on the way up, one hand-off on the way down (DESIGN.md 4.3.1)"}
        if args.workload == "c4" and not sharded and args.cold_copies > 1:
            # The same step with the operands coming from HBM: COPIES contexts of the same program
            # are cycled, so a context's 64 MB of A matrices are evicted from the 256 MiB Infinity
            # Cache (and from L2) before their next use.
            copies = [ctx]
            for _ in range(args.cold_copies - 1):
                c2 = syn.build(KktContext, prob, "lmi", device=local_rank, stream=stream)
                for i in range(c2.K):
                    c2.set_W(i, W[i])
                c2.set_cost(prob["b"])
                copies.append(c2)
            for c2 in copies:
                c2.kkt_solve_async(0.7, 0.9, 0.8)
            for c2 in copies:
                c2.sync()
            ctx.enable_timing(1)
            ctx.kernel_clock("assembly", reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = max(1, args.steps // 4)
            for r in range(reps):
                for c2 in copies:
                    c2.kkt_solve_async(0.7, 0.9, 0.8)
            for c2 in copies:
                c2.sync()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            ctx.enable_timing(False)
            ns2, km2 = ctx.kernel_clock("assembly", reset=True)
            out["cold_cache"] = {"copies": args.cold_copies,
                                 "resident_bytes_cycled": args.cold_copies * abytes,
                                 "value": reps * len(copies) / el, "unit": "KKT-solves/s",
                                 "ms_per_step": 1e3 * el / (reps * len(copies)),
                                 "kernel_ms": km2, "kernel_samples": ns2,
                                 "achieved_GBps": (abytes / (km2 * 1e-3) / 1e9) if km2 > 0 else None}
            if km2 > 0 and "roofline" in out:
                # both fractions side by side: `frac` with A (64 MB) resident in the 256 MiB Infinity
                # Cache as a back-to-back loop leaves it, `frac_cold_cache` with the operands from DRAM
                out["roofline"]["frac_cold_cache"] = abytes / (km2 * 1e-3) / 1e9 / HBM_PEAK_GBS
            del copies
        if args.workload == "c4" and not sharded and not args.no_newton_step and args.K == 1000:
            try:
                out["newton_step"] = newton_step_section(prob, local_rank, {"slab_bytes": slab_bytes, "N": ctx.N})
            except Exception as e:   # the headline line must not depend on the secondary metric
                out["newton_step"] = {"error": repr(e)}
        if not args.no_cpu and not sharded:
            cb, yo = cpu_baseline(prob, W, kind=kind)
            out["cpu_baseline"] = cb
            out["config"]["direction_rel_err_vs_cpu"] = float(
                np.linalg.norm(y - yo) / np.linalg.norm(yo))
        elif not args.no_cpu:
            out["cpu_baseline"] = None
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
