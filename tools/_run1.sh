cd /root/repo
python -m pytest tests/test_kernels_gpu.py -x -q -k "layernorm_window_modes" 2>&1 | tail -3
