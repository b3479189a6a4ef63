#!/bin/bash
# samples the shader clock / socket power while one kernel variant runs in a loop (NMH_C48_DBG picks the conv48 diagnostic variant)
python - <<'PY' &
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from nerf_mae_amd import ops
from tests.test_kernels_gpu import _pack_via_kernel
x = torch.randn(4, 160, 160, 160, 48, device='cuda').to(torch.bfloat16)
wk = _pack_via_kernel(torch.randn(48, 48, 3, 3, 3) * (27 * 48) ** -0.5, 6, torch.bfloat16, 41 * 3 * 64 * 8)
y = torch.empty_like(x)
t0 = time.time()
while time.time() - t0 < 8:
    for _ in range(50): ops.conv3d_k3_c48(x, wk, out=y)
    torch.cuda.synchronize()
PY
PID=$!
sleep 5
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -4; sleep 0.5; done
wait $PID
