OUT=$1; shift
: > $OUT
for rep in 1 2 3 4 5; do
  for e in "$@"; do
    v=$(env $e python bench.py --no-cpu-baseline --no-sweep --no-kernel-timing --steps 30 --warmup 5 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')
    echo "rep $rep  [$e]  $v" >> $OUT
  done
done
