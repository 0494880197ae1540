#!/bin/bash
# usage: scratch/ab.sh "<sono_bench args>" variant...   (variant "cur" = the in-tree library); interleaved, 3 rounds
args=$1; shift
for r in 1 2 3; do
  for v in "$@"; do
    if [ $v = cur ]; then L=""; else L=$PWD/scratch/$v; fi
    echo -n "$v: "; LD_LIBRARY_PATH=$L timeout -k 10 60 ./tools/sono_bench $args | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['route'])" || exit 1
  done
done
