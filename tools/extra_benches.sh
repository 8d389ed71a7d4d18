#!/bin/bash
# The other BASELINE configurations and side workloads, one bench.py line each (run on the GPU box):
#   tools/extra_benches.sh r02      -> gpurun_out/<round>/extra/bench_<workload and flags>.json
# Copy what should be judged into profiles/<round>/extra/.
ROUND=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$ROUND/extra
mkdir -p "$OUT"
cd "$ROOT"
for spec in "c2" "c3" "c3 --soc-tree 8" "c4s" "c5" "maxcut" "maxcut --maxcut-n 500"; do
  name=$(echo "bench_$spec" | tr ' -' '__')
  python3 bench.py --workload $spec > "$OUT/$name.json" 2> "$OUT/$name.err"
  tail -c 300 "$OUT/$name.json" | head -c 0
  python3 - "$OUT/$name.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-70s %10.1f %s  (%.1f us/step)" % (d["config"]["workload"][:70], d["value"], d["unit"], 1e3 * d["ms_per_step"]))
PY
done
