#!/usr/bin/env python3
"""profiles/<round>/gemm_*: the isolated GEMM rates, the rocprofv3 kernel statistics of the same
script and, per dispatch, duration next to the MFMA / VALU counters of the PMC passes
(tools/gemm_rocprof.sh writes gpurun_out/<round>/gemm/).

MFMA utilisation of a dispatch = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CYCLES / ... ) is
not formed here (the counters' normalisation on gfx950 is not documented); the table gives the
raw counters and  mfma_cycles_per_us = SQ_VALU_MFMA_BUSY_CYCLES / duration  so that dispatches can
be compared with each other and with the pure-MFMA microbenchmark."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", rnd, "gemm")
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "gemm_rates.jsonl"), os.path.join(dst, "gemm_rates.jsonl"))
st = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(dst, "gemm_kernel_stats.csv"))
rows = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if "gemm_f64" not in r["Kernel_Name"]:
            continue
        key = (os.path.basename(os.path.dirname(f)) if False else "", int(r["Dispatch_Id"]))
        d = rows.setdefault((f.split("pmc_")[1][0], int(r["Dispatch_Id"])), {
            "kernel": r["Kernel_Name"].split("(")[0].replace("void cxk::", ""), "grid": int(r["Grid_Size"]),
            "lds_bytes": int(r["LDS_Block_Size"]), "vgprs": int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]),
            "duration_us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
# the two passes run the same dispatch sequence: merge by dispatch id
merged = {}
for (p, did), d in rows.items():
    m = merged.setdefault(did, {})
    for k, v in d.items():
        if k == "duration_us":
            m.setdefault("duration_us", []).append(v)
        else:
            m[k] = v
names = ["dispatch", "kernel", "grid", "lds_bytes", "vgprs", "duration_us", "SQ_VALU_MFMA_BUSY_CYCLES",
         "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS",
         "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "mfma_busy_cycles_per_us", "mfma_pipe_busy_frac_at_2.4GHz"]
with open(os.path.join(dst, "gemm_dispatches.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(names)
    for did in sorted(merged):
        m = merged[did]
        dur = sum(m["duration_us"]) / len(m["duration_us"])
        w.writerow([did, m.get("kernel"), m.get("grid"), m.get("lds_bytes"), m.get("vgprs"), "%.2f" % dur] +
                   [m.get(c, "") for c in names[6:14]] +
                   ["%.0f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / dur) if "SQ_VALU_MFMA_BUSY_CYCLES" in m else "",
                    # 1024 SIMDs x 2400 cycles per microsecond: the share of the chip's fp64 matrix-pipe
                    # cycles this dispatch kept busy (the clock under load is lower, so this reads low)
                    "%.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur * 2400.0 * 1024)) if "SQ_VALU_MFMA_BUSY_CYCLES" in m else ""])
print("wrote", dst, [n for n in sorted(os.listdir(dst)) if n.startswith("gemm")])
