#!/bin/bash
# rocprofv3 evidence for the fp64 MFMA GEMM at the supernode shapes (run on the GPU box):
#   tools/gemm_rocprof.sh r02
# kernel-trace statistics of tools/gemm_profile.py --quick, then PMC passes (MFMA busy cycles,
# VALU instructions, busy cycles) in runs of their own (never combined with other tracing).
ROUND=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$ROUND/gemm
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/tools/gemm_profile.py" > "$OUT/gemm_rates.jsonl" 2> "$OUT/gemm_rates.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o gemm -- \
  python3 "$ROOT/tools/gemm_profile.py" --quick > "$OUT/under_rocprof.log" 2>&1
i=0
for SET in "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pmc_$i" -o p -- \
    python3 "$ROOT/tools/gemm_profile.py" --quick > "$OUT/pmc_$i.log" 2>&1
done
find "$OUT" -name "*.csv" | head -20
