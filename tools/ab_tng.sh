for cfg in "NMH_TNG_REG=0 NMH_TNG_BIG=0" "NMH_TNG_REG=0 NMH_TNG_BIG=512" "NMH_TNG_REG=1 NMH_TNG_BIG=0" "NMH_TNG_REG=1 NMH_TNG_BIG=512" "NMH_TNG_REG=0 NMH_TNG_BIG=256"; do
  for b in 8 1; do
    r=$(env $cfg python bench.py --no-cpu-baseline --no-kernel-timing --no-sweep --batch-per-gpu $b --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$cfg grids=$b: $r"
  done
done
