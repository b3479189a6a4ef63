"""tail_fwd_coarse_kernel alone at 8 x 160^3 (GPU box): python tools/bench_tail_coarse.py [B] [R] -- HIP-event time of the launch, best of 5 after a warm-up."""
import sys, torch
sys.path.insert(0, "/root/repo")
from nerf_mae_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 160
dt, Cd, gd, V = torch.bfloat16, 48, R // 4, R ** 3
torch.manual_seed(0)
y = (torch.randn(B * V, Cd, device="cuda") * 1.3 + 0.2).to(dt)
xc = torch.randn(B, gd, gd, gd, 96, device="cuda").to(dt)
ws = torch.empty(ops.cconv_pack_ws_floats(), dtype=torch.float32, device="cuda")
Wt, bt = torch.randn(96, 48, 4, 4, 4, device="cuda") * 96 ** -0.5, torch.randn(48, device="cuda") * 0.5
ops.cconv_pack(Wt, torch.randn(48, 48, 3, 3, 3, device="cuda"), bt, torch.empty(ops.cconv_pack_numel(), dtype=dt, device="cuda"), torch.empty(27, 48, device="cuda"), ws)
Wres = torch.empty(ops.tail_residual_pack_numel(), dtype=dt, device="cuda")
ops.tail_residual_pack(ws, Wres)
stats = torch.empty(B, Cd, 2, device="cuda")
ops.instnorm_stats(y, stats, torch.empty(B, Cd, 2, dtype=torch.float64, device="cuda"), B, V, Cd)
Wo, bo = torch.randn(4, Cd, device="cuda") * 0.2, torch.randn(4, device="cuda") * 0.1
x = torch.rand(B, 4, R, R, R, device="cuda")
ext = torch.tensor([[R, R, R]] * B, dtype=torch.int32, device="cuda")
tm = (torch.rand(gd ** 3, device="cuda") < 0.75).to(torch.uint8)
lsums, losses, dpred = torch.empty(8, dtype=torch.float64, device="cuda"), torch.empty(3, device="cuda"), torch.empty(B * V, 4, device="cuda")
bsum = torch.empty(B * Cd * 4 + 4 * Cd, dtype=torch.float64, device="cuda")
smask = torch.empty(B * V, 8, dtype=torch.uint8, device="cuda")
best = 1e9
for i in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.mae_tail_fwd_from_coarse(y, stats, xc, Wres, bt, Wo, bo, x, ext, tm, B, R, Cd, lsums, losses, dpred, bsum, smask)
    e1.record()
    torch.cuda.synchronize()
    if i:
        best = min(best, e0.elapsed_time(e1))
gb = (y.numel() * 2 + xc.numel() * 2 + B * V * (16 + 16 + 8)) / 1e9
print(f"tail_fwd_from_coarse B={B} R={R}: {best * 1e3:.0f} us incl. the zeroing / finalize launches, {gb:.2f} GB algorithmic -> {gb / best:.2f} TB/s")
