#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_profile.sh <tag> <grids per GPU> [extra bench.py flags]
# rocprofv3 kernel trace + stats of a short bench run -> gpurun_out/<tag>/{stats/, step_shapes.txt, b.log}; the (large) trace CSV is dropped.
set -u
TAG=$1; BPG=$2; shift 2
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench -- python "$REPO/bench.py" --batch-per-gpu "$BPG" --no-cpu-baseline --no-kernel-timing --no-sweep --steps 12 --warmup 3 "$@" > "$OUT/b.log" 2>&1
cd "$REPO"
TR=$(find "$OUT" -name '*kernel_trace.csv' | head -1)
if [ -n "$TR" ]; then
  python tools/trace_shapes.py "$OUT" 400 > "$OUT/step_shapes.txt" 2>> "$OUT/b.log"
  mkdir -p "$OUT/stats"
  find "$OUT" -name '*kernel_stats.csv' -exec cp {} "$OUT/stats/bench_kernel_stats.csv" \;
  find "$OUT" -mindepth 1 -maxdepth 1 -type d ! -name stats -exec rm -rf {} +
  find "$OUT" -name '*kernel_trace.csv' -delete
else
  echo "no kernel trace produced" >> "$OUT/b.log"
fi
grep -h '"metric"' "$OUT/b.log" | cut -c1-200
head -3 "$OUT/step_shapes.txt"
