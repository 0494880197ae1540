#!/bin/bash
# Kernel-trace statistics of the other entry points of the path (native host, 600 s of audio each):
# plugin getFrequencies (table twiddles), per-frame features, 512 / 2048 / 4096 points.
# Usage (repo root, on the box): bash tools/profile_variants.sh <tag>
set -u
TAG=${1:-variants}
OUT=gpurun_out/prof_$TAG
W=/tmp/prof_$TAG
mkdir -p "$OUT" "$W"
export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$W/$name" -o t -- ./tools/sono_bench "$@" > "$OUT/$name.log" 2>&1 || { echo "$name failed"; tail -3 "$OUT/$name.log"; exit 1; }
  find "$W/$name" -name "*kernel_stats.csv" -exec cp {} "$OUT/${name}_kernel_stats.csv" \;
  tail -1 "$OUT/$name.log" > "$OUT/$name.json"
}
run plugin_u8_1024 600 5 u8 1024 256 plugin
run plugin_features_1024 600 5 feat 1024 256 plugin
run worker_u8_512 600 5 u8 512 128 worker
run worker_u8_2048 600 5 u8 2048 512 worker
run plugin_u8_2048 600 5 u8 2048 512 plugin
run plugin_u8_4096 600 5 u8 4096 256 plugin
run plugin_rgba_4096_crop 60 5 rgba 4096 256 plugin 160 1280
run plugin_features_512 600 5 feat 512 128 plugin
run plugin_features_4096 600 5 feat 4096 1024 plugin
run worker_rgba_512 600 5 rgba 512 128 worker
run worker_rgba_2048 600 5 rgba 2048 512 worker
run plugin_rgba_2048 600 5 rgba 2048 512 plugin
run plugin_u8_1024_exact 120 3 u8 1024 256 plugin 0 0 exact
run plugin_u8_1024_mel 600 5 u8 1024 256 plugin 0 0 fast mel
run worker_rgba_1024_exact 120 3 rgba 1024 256 worker 0 0 exact
run plugin_rgba_1024_mel 600 5 rgba 1024 256 plugin 0 0 fast mel
run plugin_rgba_1024 600 5 rgba 1024 256 plugin
run plugin_u8_512 600 5 u8 512 128 plugin
run plugin_rgba_512 600 5 rgba 512 128 plugin
run plugin_u8_512_exact 120 3 u8 512 128 plugin 0 0 exact
run plugin_u8_2048_exact 120 3 u8 2048 512 plugin 0 0 exact
run plugin_u8_4096_exact 120 3 u8 4096 1024 plugin 0 0 exact
run worker_rgba_2048_exact 120 3 rgba 2048 512 worker 0 0 exact
export SONO_BENCH_WINDOW=8   # rectangular window: a lower loudness threshold under the plugin's table
run plugin_u8_1024_rectangular 600 5 u8 1024 256 plugin
unset SONO_BENCH_WINDOW
# the same recording with calls of 3 % of full scale instead of 30 %: no unit is loud, nothing is transformed twice
export SONO_BENCH_PULSE=0.03
run quiet_plugin_u8_1024 600 5 u8 1024 256 plugin
run quiet_plugin_rgba_1024 600 5 rgba 1024 256 plugin
run quiet_plugin_u8_512 600 5 u8 512 128 plugin
run quiet_plugin_u8_2048 600 5 u8 2048 512 plugin
run quiet_plugin_u8_4096 600 5 u8 4096 256 plugin
run quiet_plugin_rgba_4096_crop 60 5 rgba 4096 256 plugin 160 1280
unset SONO_BENCH_PULSE
rm -f "$OUT"/*.log
ls "$OUT"
