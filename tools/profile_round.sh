#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun):
#   tools/profile_round.sh r01
# kernel-trace statistics of the default bench.py run, then one PMC pass per counter set (PMC
# passes never share a run with other tracing, MI355X_MICROARCH.md HBM/rocprofv3 section).
# Output lands in gpurun_out/<round>/; tools/summarize_profiles.py condenses it into profiles/.
set -u
ROUND=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$ROUND
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
# (the driver's command line)
python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_settings.json" 2> "$OUT/bench_driver_settings.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- \
  python3 "$ROOT/bench.py" --no-cpu > "$OUT/bench_default_under_rocprof.log" 2>&1
python3 "$ROOT/bench.py" --workload c2 > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c2" -o bench -- \
  python3 "$ROOT/bench.py" --workload c2 --no-cpu --steps 50 > "$OUT/bench_c2_under_rocprof.log" 2>&1
# timeline of one steady-state step; the in-kernel timeline of the whole-tree launch (diagnostic
# library: make -C conex_amd/csrc dbg); the interior-point loop through conex.h, wall and per kernel
"$ROOT/tools/step_timeline.sh" c4 > "$OUT/step_timeline.txt" 2>&1
CXK_NO_FUSED_TREE=1 "$ROOT/tools/step_timeline.sh" c4_levels > "$OUT/step_timeline_level_kernels.txt" 2>&1
python3 "$ROOT/tools/fused_tree_stamps.py" > "$OUT/fused_tree_stamps.txt" 2>&1
python3 "$ROOT/tools/fused_tree_stamps.py" 1 200 50 > "$OUT/fused_tree_stamps_c2.txt" 2>&1
# big supernodes: in-kernel timeline of big_chol_dataflow at 500 columns, kernel statistics of the max-cut step
python3 "$ROOT/tools/big_chol_stamps.py" 500 > "$OUT/big_chol_stamps.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_maxcut500" -o bench -- \
  python3 "$ROOT/bench.py" --workload maxcut --maxcut-n 500 --no-cpu --steps 100 > "$OUT/bench_maxcut500_under_rocprof.log" 2>&1
# sharding, rehearsed on the one GPU (rank 0 of a virtual world, one-rank RCCL communicator standing in for
# the exchange): the sharded step on the whole-tree kernels and on the level kernels, C4 and C5
: > "$OUT/shard_rehearsal.jsonl"
for W in 2 4 8; do
  for WL in c4 c5; do
    python3 "$ROOT/bench.py" --workload $WL --shard-path --shard-world $W --no-cpu 2>/dev/null | tail -1 >> "$OUT/shard_rehearsal.jsonl"
    CXK_NO_FUSED_SHARD=1 python3 "$ROOT/bench.py" --workload $WL --shard-path --shard-world $W --no-cpu 2>/dev/null | tail -1 >> "$OUT/shard_rehearsal.jsonl"
  done
done
# isolated rates of the fp64 MFMA GEMM at the supernode shapes of SURVEY 8(d)
python3 "$ROOT/tools/gemm_profile.py" > "$OUT/gemm_rates.jsonl" 2> "$OUT/gemm_rates.err"
"$ROOT/tools/extra_benches.sh" "$ROUND" > "$OUT/extra_benches.txt" 2>&1
python3 "$ROOT/tools/ipm_iteration.py" --timers > "$OUT/ipm_iteration.txt" 2>&1
# (the timed run takes the path with a host round trip per phase; this one is the product's)
python3 "$ROOT/tools/ipm_iteration.py" > "$OUT/ipm_iteration_wall.txt" 2>&1
"$ROOT/tools/ipm_rocprof.sh" 2>&1 | grep -v "^[EW]20" > "$OUT/ipm_kernels.txt"
cd /tmp
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pmc_$i" -o p -- \
    python3 "$ROOT/bench.py" --steps 20 --warmup 3 --no-cpu > "$OUT/pmc_$i.log" 2>&1
done
ls -R "$OUT" | head -60
