#!/bin/bash
# Profiles the hot path on the GPU box.
#   1. rocprofv3 --kernel-trace --stats over `python3 bench.py` (the benchmark command itself)
#   2. PMC passes, one run each, over the native host tools/sono_bench (same workload, no PyTorch)
# Only CSV summaries are kept: gpurun_out/prof_<tag>/ ; tools/profile_summary.py turns them into
# summary.json and traffic.json.
# Usage (repo root, on the box): bash tools/profile.sh <tag> [seconds=3600]
set -u
TAG=${1:-profile}
SEC=${2:-3600}
OUT=gpurun_out/prof_$TAG
W=/tmp/prof_$TAG
mkdir -p "$OUT" "$W"
export TMPDIR=/tmp
echo "[profile] kernel trace of bench.py"
timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$W/trace" -o trace -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --seconds $SEC > "$OUT/trace_bench.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace_bench.log"; exit 1; }
find "$W/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \;
grep '^{"metric"' "$OUT/trace_bench.log" | tail -1 > "$OUT/bench_under_rocprof.json"
pass() {
  kind=$1; name=$2; shift; shift
  echo "[profile] pmc pass $kind/$name: $*"
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$W/${kind}_$name" -o pmc -- ./tools/sono_bench $SEC 3 $kind > "$OUT/${kind}_$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/${kind}_$name.log"; exit 1; }
  find "$W/${kind}_$name" -name "*counter_collection.csv" -exec cp {} "$OUT/${kind}_$name.csv" \;
}
for kind in rgba u8; do
  pass $kind sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
  pass $kind sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
  pass $kind fetch FETCH_SIZE
  pass $kind write WRITE_SIZE
  pass $kind ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
  pass $kind l2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_WRITEBACK_sum
  pass $kind tcp TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
  pass $kind grbm GRBM_GUI_ACTIVE
done
python3 tools/profile_summary.py "$OUT" $SEC
ls "$OUT"
