#!/bin/bash
# Timeline of one steady-state KKT solve of bench.py under rocprofv3 (run on the GPU box):
#   tools/step_timeline.sh <name> [bench.py arguments]
# Prints, for the kernels of one step in the middle of the timed loop, start offset, duration and
# the idle gap before each (us), plus the per-kernel averages over the run.
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
name=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 100 "$@" > $OUT/$name.log 2>&1
python3 - <<PY
import csv,glob
f=[x for x in glob.glob("$OUT/$name/**/*kernel_trace.csv",recursive=True)][0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
# a step starts at each Schur kernel
idx=[i for i,n in enumerate(names) if "lmi_schur" in n]
mid=idx[len(idx)//2]; nxt=idx[len(idx)//2+1]
t0=int(rows[mid]["Start_Timestamp"])
prev_end=None
print("== one step (kernel, start us, duration us, gap before us)")
for r in rows[mid:nxt+1]:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    gap=(s-prev_end)/1e3 if prev_end else 0.0
    print("  %-48s %8.2f %7.2f %6.2f"%(r["Kernel_Name"][:48],(s-t0)/1e3,(e-s)/1e3,gap))
    prev_end=e
import collections
d=collections.defaultdict(list)
for i in range(idx[10], idx[-2]):
    r=rows[i]; d[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
print("== averages")
for k,v in d.items(): print("  %-60s n %5d avg %7.2f us"%(k,len(v),sum(v)/len(v)))
steps=[(int(rows[idx[i+1]]["Start_Timestamp"])-int(rows[idx[i]]["Start_Timestamp"]))/1e3 for i in range(10,len(idx)-2)]
print("== step period avg %.2f us"%(sum(steps)/len(steps)))
PY
