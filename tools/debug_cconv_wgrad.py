"""debug: G blocks and the chain rule of csrc/cconv.hip's composed weight gradient against a direct torch evaluation"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from nerf_mae_amd import ops
torch.manual_seed(0)
B, v = 1, 8
Fv = 4 * v
dt = torch.bfloat16
x = torch.randn(B, v, v, v, 96).to(dt).float()
dy = torch.randn(B, Fv, Fv, Fv, 48)
dy = (dy - dy.mean(dim=(1, 2, 3), keepdim=True)).to(dt).float()
Wt = torch.randn(96, 48, 4, 4, 4) * 96 ** -0.5
W1 = torch.randn(48, 48, 3, 3, 3) * (27 * 48) ** -0.5
bt = torch.randn(48) * 0.5
def ncnt(a): return 2 if a in (0, 3) else 1
def nfirst(a): return -1 if a == 0 else 0
BLK = [(0, -1), (0, 0), (1, 0), (2, 0), (3, 0), (3, 1)]
xp = F.pad(x, (0, 0, 1, 1, 1, 1, 1, 1))      # zero cells around the coarse grid
Gref = []
for gi in range(16):
    az, ay = gi >> 2, gi & 3
    for iz in range(ncnt(az)):
        for iy in range(ncnt(ay)):
            nz, ny = nfirst(az) + iz, nfirst(ay) + iy
            for ax, nx in BLK:
                xs = xp[:, 1 + nz:1 + nz + v, 1 + ny:1 + ny + v, 1 + nx:1 + nx + v]          # x[j + n]
                dys = dy[:, az::4, ay::4, ax::4]                                              # dy[4j + a]
                Gref.append(torch.einsum("bzyxi,bzyxc->ic", xs, dys))
Gref = torch.stack(Gref)      # [216][96][48]
dev = lambda t, d=None: (t if d is None else t.to(d)).cuda().contiguous()
Wcp = torch.empty(ops.cconv_pack_numel(), dtype=dt, device="cuda"); delta = torch.empty(27, 48, device="cuda")
pws = torch.empty(ops.cconv_pack_ws_floats(), device="cuda")
ops.cconv_pack(dev(Wt), dev(W1), dev(bt), Wcp, delta, pws)
dW = torch.zeros(48, 48, 3, 3, 3, device="cuda")
ops.cconv_wgrad(dev(x, dt), dev(dy, dt), pws, dev(bt), dW, B, v)
torch.cuda.synchronize()
ws = list(ops._CCW_WS.values())[0]
G = ws[256 * 55296:256 * 55296 + 216 * 4608].view(216, 96, 48).cpu()
err = (G - Gref).abs().amax(dim=(1, 2)) / Gref.abs().amax()
print("G blocks: worst rel err", err.max().item(), "bad blocks:", [i for i in range(216) if err[i] > 1e-2][:40])
# chain rule from the reference G
W1r = W1.clone().requires_grad_(True)
u = F.conv_transpose3d(x.permute(0, 4, 1, 2, 3), Wt, bt, stride=4)
(F.conv3d(u, W1r, None, padding=1) * dy.permute(0, 4, 1, 2, 3)).sum().backward()
print("dW vs autograd:", ((dW.cpu() - W1r.grad).abs().max() / W1r.grad.abs().max()).item())
Cb = ws[256 * 55296 + 216 * 4608:256 * 55296 + 216 * 4608 + 27 * 48].view(27, 48).cpu()
cls = torch.ones(Fv, dtype=torch.long); cls[0] = 0; cls[-1] = 2
cid = (cls[:, None, None] * 3 + cls[None, :, None]) * 3 + cls[None, None, :]
Cref = torch.zeros(27, 48)
for k in range(27):
    if k != 13: Cref[k] = dy[0][cid == k].sum(0)
print("border class sums:", ((Cb - Cref).abs().max() / Cref.abs().max()).item())
blk = None
base = 0
for gi in range(16):
    if gi == 5:
        blk = base * 6 + 2
    base += ncnt(gi >> 2) * ncnt(gi & 3)
print("block", blk)
torch.set_printoptions(precision=3, linewidth=200)
print(G[blk][:6, :6]); print(Gref[blk][:6, :6])
print("corr of flattened:", torch.corrcoef(torch.stack([G[blk].flatten(), Gref[blk].flatten()]))[0, 1].item())
print("G[blk] vs Gref[blk].T-ish shapes", G[blk].shape)
# is G some other block of Gref?
flat = Gref.view(216, -1)
for cand in (G[blk].flatten(),):
    sims = (flat @ cand) / (flat.norm(dim=1) * cand.norm() + 1e-9)
    print("best matching ref block", sims.argmax().item(), sims.max().item())
# per-row / per-col norms ratio
print("row-norm ratio", (G[blk].norm(dim=1) / Gref[blk].norm(dim=1))[:12])
print("col-norm ratio", (G[blk].norm(dim=0) / Gref[blk].norm(dim=0))[:12])
part = ws[:256 * 55296].view(256, 2, 6, 96, 48).cpu()
print("partials abs sum per wg (first 40):", [round(v, 1) for v in part.abs().sum(dim=(1, 2, 3, 4))[:40].tolist()])
print("G abs sum", G.abs().sum().item())
print("G block sums:", [int(v) for v in G.abs().sum(dim=(1, 2)).tolist()])
print("ref block sums:", [int(v) for v in Gref.abs().sum(dim=(1, 2)).tolist()])
# replicate the host's unit table
units = []; base = 0
for gi in range(16):
    az, ay = gi >> 2, gi & 3
    cz, cy = ncnt(az), ncnt(ay)
    nc = cz * cy
    for first in range(0, nc, 2):
        units.append(dict(gi=gi, ncomb=2 if nc - first >= 2 else 1, blk0=base + first, blk1=base + first + 1))
    base += nc
wsum = sum(30.7 + 16.1 * u["ncomb"] for u in units)
wg = 0
npair = B * v * (v // 2)
for u in units:
    s = max(1, min(int(248.0 * (30.7 + 16.1 * u["ncomb"]) / wsum), npair))
    u["wg0"], u["nslab"] = wg, s; wg += s
print("total wgs", wg)
def rel(a, b): return ((a - b).abs().max() / b.abs().max()).item()
for u in units[:6]:
    tot = part[u["wg0"]:u["wg0"] + u["nslab"]].sum(0)      # [2][6][96][48]
    r00 = rel(tot[0], Gref[u["blk0"] * 6:u["blk0"] * 6 + 6])
    r01 = rel(tot[0], Gref[u["blk1"] * 6:u["blk1"] * 6 + 6]) if u["ncomb"] > 1 else -1
    r11 = rel(tot[1], Gref[u["blk1"] * 6:u["blk1"] * 6 + 6]) if u["ncomb"] > 1 else -1
    r10 = rel(tot[1], Gref[u["blk0"] * 6:u["blk0"] * 6 + 6]) if u["ncomb"] > 1 else -1
    print(u, "slot0~blk0 %.3g slot0~blk1 %.3g slot1~blk1 %.3g slot1~blk0 %.3g" % (r00, r01, r11, r10), "slot1 abs", tot[1].abs().sum().item())
Gem = torch.zeros(216, 96, 48)
for u in units:
    tot = part[u["wg0"]:u["wg0"] + u["nslab"]].sum(0)
    Gem[u["blk0"] * 6:u["blk0"] * 6 + 6] = tot[0]
    if u["ncomb"] > 1:
        Gem[u["blk1"] * 6:u["blk1"] * 6 + 6] = tot[1]
print("emulated reduce vs ref:", rel(Gem, Gref), " GPU G vs emulated:", rel(G, Gem))
print("units:", [(u["blk0"], u["blk1"], u["ncomb"], u["wg0"], u["nslab"]) for u in units])
