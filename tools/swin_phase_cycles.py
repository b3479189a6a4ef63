"""phase timestamps of the fused Swin-block forward kernels (NMH_SWIN_DBG=8 builds record s_memtime at phase boundaries, wave 0 of workgroup 0;
the launcher prints the differences to stderr).  usage: NMH_SWIN_DBG=8 python tools/swin_phase_cycles.py [grids = 8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NMH_SWIN_DBG", "8")
import torch
from nerf_mae_amd import ops

grids = int(sys.argv[1]) if len(sys.argv) > 1 else 8
C, s = 384, 10
dt = torch.bfloat16
M = s ** 3 * grids
x = torch.randn(M, C, device="cuda").to(dt)
gam, bet = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
Wqkv = torch.randn(3 * C, C, device="cuda") * C ** -0.5; Wp = torch.randn(C, C, device="cuda") * C ** -0.5
W1 = torch.randn(4 * C, C, device="cuda") * C ** -0.5; W2 = torch.randn(C, 4 * C, device="cuda") * (4 * C) ** -0.5
z = lambda n: torch.zeros(n, device="cuda")
table = torch.randn(343, C // 32, device="cuda") * 0.02
geom = ops.WinGeom(grids, s, s, s, [2, 2, 2])
st = {k: torch.empty(ops.swin_stream_numel(k, C), dtype=dt, device="cuda") for k in (0, 1)}
ops.swin_pack(ops.swin_pack_items([(Wqkv, Wp, st[0], 0, C), (W1, W2, st[1], 1, C)]))
for _ in range(4):
    ops.swin_attn_fwd(x, gam, bet, st[0], z(3 * C), table, z(C), geom)
    ops.swin_mlp_fwd(x, gam, bet, st[1], z(4 * C), z(C))
torch.cuda.synchronize()
