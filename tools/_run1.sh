cd /root/repo
python -m pytest tests/test_kernels_gpu.py tests/test_surface_gpu.py -x -q -k "attention or attn" 2>&1 | tail -2
for g in 8 8; do python bench.py --batch-per-gpu $g --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-sweep 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print($g, d['ms_per_step'], d['value'])"; done
