#!/bin/bash
# Profiles the hot path on the GPU box.
#   1. rocprofv3 --kernel-trace --stats over `python3 bench.py` (the benchmark command itself)
#   2. PMC passes, one run each, over the native host tools/sono_bench (few dispatches, no PyTorch)
# Only the CSV summaries are copied to gpurun_out/prof_<tag>/.
# Usage (repo root, on the box): bash tools/profile.sh <tag> [out=u8|rgba]
set -u
TAG=${1:-r01}
OUTK=${2:-u8}
OUT=gpurun_out/prof_$TAG
W=/tmp/prof_$TAG
mkdir -p "$OUT" "$W"
export TMPDIR=/tmp
echo "[profile] kernel trace of bench.py"
timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$W/trace" -o trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variants --out $OUTK > "$OUT/trace_bench.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace_bench.log"; exit 1; }
find "$W/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \;
pass() {
  name=$1; shift
  echo "[profile] pmc pass $name: $*"
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$W/$name" -o pmc -- ./tools/sono_bench 600 3 $OUTK > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
  find "$W/$name" -name "*counter_collection.csv" -exec cp {} "$OUT/$name.csv" \;
}
pass pmc_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
pass pmc_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU
pass pmc_fetch FETCH_SIZE
pass pmc_write WRITE_SIZE
pass pmc_grbm GRBM_GUI_ACTIVE
ls -la "$OUT"
