cd /root/repo
run() { python bench.py --batch-per-gpu $1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-sweep 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', $1, d['ms_per_step'], d['value'])"; }
for g in 1 2; do
  run $g base
  for mw in 64 128 192 256; do NMH_DEFER_W48=1 NMH_W48_MAXWG=$mw run $g "defer_maxwg$mw"; done
done
