# same-box A/B of environment knobs on the 8-grid bench step (two alternating rounds); usage: bash tools/ab_knobs.sh [grids] "K1=V1" "K2=V2" ...
G=${1:-8}; shift
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-sweep --global-batch $G"
run() { echo -n "$1: "; env $1 $B 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
for r in 1 2 3; do
  run X=0
  for kv in "$@"; do run "$kv"; done
done
