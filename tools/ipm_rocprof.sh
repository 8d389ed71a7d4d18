#!/bin/bash
# Per-kernel device time of CONEX_Maximize on C4 (three solves) under rocprofv3; run on the GPU box.
OUT=$GRAFT_REPO_ROOT/gpurun_out/ipm; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/ipm_iteration.py > $OUT/run.log 2>&1
tail -4 $OUT/run.log
python3 - <<PY
import csv,glob,re
f=glob.glob("$OUT/trace/**/*kernel_stats.csv",recursive=True)[0]
log=open("$OUT/run.log").read()
its=sum(int(x) for x in re.findall(r"iterations=(\d+)",log))
print("iterations in the run:",its)
tot=0
for r in csv.DictReader(open(f)):
    per=float(r["TotalDurationNs"])/1e3/its
    tot+=per
    if per>0.5: print("  %-64s calls/iter %5.2f  avg %7.2f us  per iteration %7.2f us"%(r["Name"][:64], int(r["Calls"])/its, float(r["AverageNs"])/1e3, per))
print("  kernel time per iteration: %.1f us"%tot)
PY
