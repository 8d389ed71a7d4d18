#!/usr/bin/env python3
"""Condense gpurun_out/<round>/ (written by tools/profile_round.sh) into profiles/<round>/.

  python tools/summarize_profiles.py r01

Copies the kernel statistics and bench lines, and averages every collected PMC counter per
kernel.  HBM traffic of the dominant kernel = 2 x FETCH_SIZE + WRITE_SIZE: on gfx950 FETCH_SIZE
counts 128-byte requests as 64 bytes (MI355X_MICROARCH.md, HBM / rocprofv3 section); both counters
are reported in KB by rocprofv3.
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def blob_hash(path):
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", rnd)
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    for name in ["bench_default.json", "bench_default_under_rocprof.log", "bench_c2.json",
                 "bench_c2_under_rocprof.log", "step_timeline.txt", "step_timeline_level_kernels.txt",
                 "fused_tree_stamps.txt", "fused_tree_stamps_c2.txt", "ipm_iteration.txt", "ipm_iteration_wall.txt", "ipm_kernels.txt",
                 "bench_driver_settings.json", "big_chol_stamps.txt", "gemm_rates.jsonl", "extra_benches.txt"]:
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, name))
    for sub, out in [("stats", "bench_default_kernel_stats.csv"), ("stats_c2", "bench_c2_kernel_stats.csv"),
                     ("stats_maxcut500", "bench_maxcut500_kernel_stats.csv")]:
        files = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
        if files:
            shutil.copy(files[0], os.path.join(dst, out))
    # the sharding rehearsal, one compact record per run
    reh = os.path.join(src, "shard_rehearsal.jsonl")
    if os.path.exists(reh):
        rows = []
        for line in open(reh):
            line = line.strip()
            if not line.startswith("{"):
                continue
            d = json.loads(line)
            rows.append({"workload": d["config"]["workload"][:24], "parallelism": d["config"]["parallelism"],
                         "whole_tree_kernels": d["config"].get("fused_tree"),
                         "sharded_us_per_step": d["layouts"]["sharded_us_per_step"],
                         "unsharded_us_per_step": d["layouts"]["replicated_us_per_step"],
                         "exchange_bytes_per_solve": d["layouts"]["exchange_bytes_per_solve"],
                         "assembly_kernel_us": 1e3 * d.get("roofline", {}).get("kernel_ms", 0),
                         "tree_launches_us": 1e3 * d.get("roofline_tree", {}).get("kernel_ms", 0)})
        with open(os.path.join(dst, "shard_rehearsal.json"), "w") as f:
            json.dump({"what": "bench.py --shard-path --shard-world W on ONE GPU: this GPU is rank 0 of a virtual world of W, "
                               "its all-reduces run on a one-rank RCCL communicator (they return their input: the other "
                               "ranks' contributions are missing, so timing and plumbing are meaningful, the direction is "
                               "not); whole_tree_kernels false = CXK_NO_FUSED_SHARD=1 (level kernels + pack / unpack launches)",
                       "runs": rows}, f, indent=1)
    extra_src = os.path.join(src, "extra")
    if os.path.isdir(extra_src):
        os.makedirs(os.path.join(dst, "extra"), exist_ok=True)
        for fn in os.listdir(extra_src):
            if fn.endswith(".json"):
                shutil.copy(os.path.join(extra_src, fn), os.path.join(dst, "extra", fn))
    counters = {}
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            c = counters.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0])
            c[0] += float(r["Counter_Value"])
            c[1] += 1
    summary = {
        "command": "rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python3 bench.py "
                   "--steps 20 --warmup 3 --no-cpu (one pass per counter set, tools/profile_round.sh)",
        "units": "FETCH_SIZE / WRITE_SIZE in KB as reported; gfx950: wide coalesced reads are "
                 "under-reported 2x (MI355X_MICROARCH.md HBM section)",
        "counters": {k: {c: {"mean": v[0] / v[1], "dispatches": v[1]} for c, v in d.items()}
                     for k, d in counters.items()},
    }
    for k, d in counters.items():
        key = "lmi_schur_mfma" if "lmi_schur_mfma" in k else ("lmi_schur_fused" if "lmi_schur_fused" in k else None)
        if key and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            fetch = d["FETCH_SIZE"][0] / d["FETCH_SIZE"][1] * 1024
            write = d["WRITE_SIZE"][0] / d["WRITE_SIZE"][1] * 1024
            summary[key] = {
                "FETCH_SIZE_bytes_raw": fetch,
                "FETCH_SIZE_bytes_corrected_x2": 2 * fetch,
                "WRITE_SIZE_bytes": write,
                "hbm_traffic_bytes_per_launch": 2 * fetch + write,
                "algorithmic_bytes_per_launch": 72400000.0,
                # `git hash-object` of the kernel source these counters were collected on (bench.py
                # compares it with the source it runs on and flags a stale figure)
                "kernel_source_blob": blob_hash(os.path.join(ROOT, "conex_amd", "csrc", "lmi_fused_mfma.hip")),
            }
    with open(os.path.join(dst, "pmc_summary.json"), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    print("wrote", dst, sorted(os.listdir(dst)))


if __name__ == "__main__":
    main()
