python - <<'PY' &
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from nerf_mae_amd import ops
x = torch.randn(4, 160, 160, 160, 48, device='cuda').to(torch.bfloat16)
dy = torch.randn(4, 160, 160, 160, 48, device='cuda').to(torch.bfloat16)
dW = torch.zeros(48, 48, 3, 3, 3, device='cuda')
t0 = time.time()
while time.time() - t0 < 8:
    for _ in range(50): ops.conv3d_k3_c48_wgrad(dy, x, dW)
    torch.cuda.synchronize()
PY
PID=$!
sleep 5
for i in 1 2; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket" | head -2; sleep 0.5; done
wait $PID
