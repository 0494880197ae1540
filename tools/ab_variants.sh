#!/bin/bash
# Same-box A/B of library variants built by tools/build_variant.sh (scratch/<name>/libsonospec.so).
# usage (repo root, on the box): bash tools/ab_variants.sh "<sono_bench args>" variant...
#   variant "cur" = the in-tree library; the runs are interleaved, 3 rounds; prints ms per step, the output checksum
#   (equal checksums: the variants write the same bytes) and the kernel route.
args=$1; shift
for r in 1 2 3; do
  for v in "$@"; do
    if [ $v = cur ]; then L=""; else L=$PWD/scratch/$v; fi
    echo -n "$v: "; LD_LIBRARY_PATH=$L timeout -k 10 60 ./tools/sono_bench $args | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['checksum'], d['route'])" || exit 1
  done
done
