#!/bin/bash
# usage (GPU box, repo root): tools/ab_env3.sh <out file> "<ENV=..>" ...   as ab_env.sh with three interleaved repetitions of 30 steps
OUT=$1; shift
: > $OUT
for rep in 1 2 3; do
  for e in "$@"; do
    v=$(env $e python bench.py --no-cpu-baseline --no-sweep --no-kernel-timing --steps 30 --warmup 5 $BENCH_ARGS 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')
    echo "rep $rep  [$e]  $v" >> $OUT
  done
done
cat $OUT
