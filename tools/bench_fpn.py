"""nerf_rpn backbone (swin_s encoder + FPN(256), SURVEY 8(f) rank 1 / BASELINE config 5 shape): forward+backward grids/s at 160^3, bf16,
eager and HIP-graph replay, with the CPU oracle on one grid beside it.   python tools/bench_fpn.py [grids_per_step] [--no-cpu]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_mae_amd.fpn import SwinTransformer_FPN_Pretrained_Skip
from nerf_mae_amd import ops
from oracle import mae3d_oracle as O   # cpu baseline + synthetic inputs only

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4
torch.manual_seed(0)
m = SwinTransformer_FPN_Pretrained_Skip(resolution=160, is_eval=True).cuda().train()
x = torch.stack([O.synthetic_grid((160, 160, 160), i) for i in range(B)]).cuda()
ops.side_stream.auto(B)


def step():
    ys = m(x)
    torch.autograd.backward(ys, [gy for gy in GY])


with torch.no_grad():
    pass
ys = m(x)
GY = [torch.randn_like(y) * 1e-3 for y in ys]
torch.autograd.backward(ys, GY)
for _ in range(2): step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
a.record()
for _ in range(n): step()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / n
print(f"eager : {ms:8.2f} ms/step  {B / ms * 1e3:7.1f} grids/s  (B={B})")
try:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    a.record()
    for _ in range(n): g.replay()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    print(f"graph : {ms:8.2f} ms/step  {B / ms * 1e3:7.1f} grids/s  (B={B})")
except Exception as e:  # noqa: BLE001
    print("graph capture failed:", repr(e)[:300])
if "--no-cpu" not in sys.argv:
    import psutil
    torch.set_num_threads(psutil.cpu_count(logical=False) or 8)
    ora = O.FPNSkipOracle(resolution=160).train()
    xc = x[:1].cpu()
    t0 = time.perf_counter()
    yo = ora(xc)
    torch.autograd.backward(yo, [g[:1].cpu() for g in GY])
    dt = time.perf_counter() - t0
    print(f"cpu oracle (fp32, {torch.get_num_threads()} threads): {dt:.2f} s/grid  {1 / dt:.4f} grids/s")
