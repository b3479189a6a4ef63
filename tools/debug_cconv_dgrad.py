"""which cells of cconv_dgrad differ from the two-step HIP path (debug aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_mae_amd import ops
from tests.test_kernels_gpu import _pack_via_kernel
B, v = int(sys.argv[1]), int(sys.argv[2])
dt = torch.bfloat16
Fv = 4 * v
g = torch.Generator().manual_seed(0)
Wt = torch.randn(96, 48, 4, 4, 4, generator=g) * 96 ** -0.5
W1 = torch.randn(48, 48, 3, 3, 3, generator=g) * (27 * 48) ** -0.5
bt = torch.randn(48, generator=g)
dy = torch.randn(B, Fv, Fv, Fv, 48, generator=g).to(dt).cuda()
Wcp = torch.empty(ops.cconv_pack_numel(), dtype=dt, device="cuda"); delta = torch.empty(27, 48, device="cuda")
ops.cconv_pack(Wt.cuda(), W1.cuda(), bt.cuda(), Wcp, delta)
Wdp = torch.empty(ops.cconv_dgrad_pack_numel(), dtype=dt, device="cuda")
ops.cconv_dgrad_pack(Wcp, Wdp)
dx = ops.cconv_dgrad(dy.view(-1, 48), Wdp, B, v)
wkd = _pack_via_kernel(W1, 7, dt, 41 * 3 * 64 * 8)
dcat = ops.conv3d_k3_c48(dy, wkd)
wtd = _pack_via_kernel(Wt, 5, dt, Wt.numel())
dx2 = torch.empty(B * v ** 3, 96, dtype=dt, device="cuda")
ops.upconv_dgrad(dcat.view(-1, 48), wtd.view(96, 64 * 48), dx2, B, v, 4, 96, 48)
torch.cuda.synchronize()
e = (dx.float() - dx2.float()).abs().amax(dim=1).view(B, v, v, v).cpu()
scale = dx2.float().abs().max().item()
bad = (e > 0.05 * scale).nonzero()
print("scale", scale, "bad cells", len(bad), "of", e.numel(), "nan", int(torch.isnan(dx.float()).sum()))
if len(bad):
    print("first", bad[:10].tolist()); print("last", bad[-5:].tolist())
    for d in range(4):
        vals = sorted(set(bad[:, d].tolist())); print("dim", d, "values", vals[:40])
