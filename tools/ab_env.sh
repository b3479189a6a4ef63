#!/bin/bash
# usage (GPU box, repo root): tools/ab_env.sh <out file> "<ENV=.. ENV=..>" ...   one short bench.py run per environment setting, interleaved twice
OUT=$1; shift
: > $OUT
for rep in 1 2; do
  for e in "$@"; do
    v=$(env $e python bench.py --no-cpu-baseline --no-sweep --no-kernel-timing --steps 20 --warmup 5 $BENCH_ARGS 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')
    echo "rep $rep  [$e]  $v" >> $OUT
  done
done
cat $OUT
