cd /root/repo
python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_nt" 2>&1 | tail -3
python tools/bench_gemm_small.py 2>/dev/null
for g in 1 2 4; do python bench.py --batch-per-gpu $g --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-sweep 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print($g, d['ms_per_step'], d['value'])"; done
