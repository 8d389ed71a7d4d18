#!/bin/bash
# Per-kernel average durations of one bench.py run under rocprofv3 (run on the GPU box):
#   tools/kernel_stats.sh <name> [ENV=VAL ...] -- [bench.py arguments]
# e.g.  tools/kernel_stats.sh c5 -- --workload c5 ;  tools/kernel_stats.sh generic CXK_NO_LEAN=1 --
OUT=$GRAFT_REPO_ROOT/gpurun_out/kernel_stats; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
name=$1; shift
while [ "$1" != "--" ] && [ -n "$1" ]; do export "$1"; shift; done
shift
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 100 "$@" > $OUT/$name.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/$name/**/*kernel_stats.csv",recursive=True)[0]
print("== $name", open("$OUT/$name.log").read()[-2000:].split('"value": ')[1][:8] if '"value"' in open("$OUT/$name.log").read() else "")
for r in csv.DictReader(open(f)):
    if float(r["Percentage"])>0.7: print("  %-70s calls %6s avg %8.1f us"%(r["Name"][:70], r["Calls"], float(r["AverageNs"])/1000))
PY
