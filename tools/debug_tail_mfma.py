import sys, os, random, torch, numpy as np
sys.path.insert(0,'.')
from nerf_mae_amd import ops
from oracle import mae3d_oracle as O
torch.manual_seed(0)
B,R,Cd=2,32,48; V=R**3; dt=torch.bfloat16
x = torch.stack([O.synthetic_grid((R,R,R),3), O.synthetic_grid((R,R,R),4)]).cuda()
ext = torch.tensor([[R,R,R],[28,R,R]],dtype=torch.int32).cuda()
y=(torch.randn(B,V,Cd)*1.3+0.2).to(dt).cuda(); r=torch.randn(B,V,Cd).to(dt).cuda()
Wo=(torch.randn(4,Cd)*0.2).cuda(); bo=(torch.randn(4)*0.1).cuda()
tm=O.draw_block_mask((R//4,)*3,0.6,rng=random.Random(5)).to(torch.uint8).cuda()
stats, scratch = torch.empty(B,Cd,2,device="cuda"), torch.empty(B,Cd,2,dtype=torch.float64,device="cuda")
ops.instnorm_stats(y, stats, scratch, B, V, Cd)
def run(mf):
    os.environ["NMH_TAIL_MFMA"]=mf
    ls, lo, dp = torch.empty(8,dtype=torch.float64,device="cuda"), torch.empty(3,device="cuda"), torch.zeros(B*V,4,device="cuda")
    bs = torch.empty(B*Cd*4+4*Cd,dtype=torch.float64,device="cuda")
    pr = torch.zeros(B,4,R,R,R,device="cuda")
    ops.mae_tail_fwd(y.view(-1,Cd), stats, r.view(-1,Cd), None, Wo, bo, x, ext, tm, B,R,Cd, ls, lo, pr, dp, bwd_sums=bs)
    torch.cuda.synchronize()
    return ls.cpu(), lo.cpu(), dp.cpu(), bs.cpu(), pr.cpu()
a=run("0"); b=run("1")
print("sums", a[0].numpy(), b[0].numpy())
print("losses", a[1].numpy(), b[1].numpy())
d=(a[2]-b[2]).abs(); print("dpred maxdiff per o", d.max(0).values.numpy(), "max", a[2].abs().max().item())
bad=(d>1e-4).nonzero(); print("bad count", len(bad), bad[:12].tolist())
pd=(a[4]-b[4]).abs(); print("pred maxdiff", pd.max().item(), "per o", pd.amax(dim=(0,2,3,4)).numpy())
bsd=(a[3]-b[3]).abs()/ (a[3].abs().max()); print("bsums rel max", bsd.max().item())
print(a[3][:8].numpy(), b[3][:8].numpy())
bad=(d>1e-3).nonzero()
print("bad", len(bad), "of", d.numel())
import collections
print("by o", collections.Counter(bad[:,1].tolist()))
vv=bad[:,0] % V
print("voxel%32 histogram", sorted(collections.Counter((vv%32).tolist()).items())[:40])
pdiff=(a[4]-b[4]).abs().reshape(B,4,-1)
pb=(pdiff>1e-3).nonzero(); print("pred bad", len(pb), sorted(collections.Counter((pb[:,2]%32).tolist()).items())[:40])
