#!/bin/bash
# SQ counter passes over one tools/sono_bench configuration (one rocprofv3 run per pass).
# Usage (repo root, on the box): bash tools/pmc_kernel.sh <tag> <sono_bench args...>
set -u
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
W=/tmp/pmc_$TAG
mkdir -p "$OUT" "$W"
export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$W/$name" -o pmc -- ./tools/sono_bench $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
  find "$W/$name" -name "*counter_collection.csv" -exec cp {} "$OUT/$name.csv" \;
}
ARGS="$*"
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass grbm GRBM_GUI_ACTIVE
python3 - "$OUT" <<'PY'
import csv, sys, collections, glob, os, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob(os.path.join(out, "*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "stft" not in k: continue
        k = k.split("(")[0].split("::")[-1][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
res = {k: {c: round(v / max(1, len(disp[(k, c)])), 1) for c, v in sorted(d.items())} for k, d in acc.items()}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
