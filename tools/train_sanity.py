"""full-size training sanity run: swin_s 160^3 bf16, 4 grids/step, HIP-graph step, N optimizer steps on a fixed synthetic batch;
prints the loss trajectory (must fall monotonically-ish and stay finite)"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_mae_amd import data
from nerf_mae_amd.model import build_model, draw_block_mask
from nerf_mae_amd.trainer import FusedAdamW, GraphedTrainStep, OneCycle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
B, R = 4, 160
torch.manual_seed(0); random.seed(0)
dev = torch.device("cuda", 0)
m = build_model("swin_s", resolution=R).to(dev).train()
m.flatten_parameters()
opt = FusedAdamW(m, lr=1e-4, weight_decay=1e-3, max_grad_norm=0.1)
sched = OneCycle(1e-4, N)
exts = [(R, R, R), (R, 132, 96), (120, R, 144), (R, R, R)]
scenes = [data.synthetic_scene(exts[i], seed=i) for i in range(B)]
xb, ext = data.GridBatcher(R, dev, normalize_density=True)(scenes, flags=[0] * B)
grids = [xb[i, :, :e[0], :e[1], :e[2]].contiguous() for i, e in enumerate(ext.tolist())]
rng = random.Random(1)
step = GraphedTrainStep(m, opt, B)
step(grids, draw_block_mask((40, 40, 40), 0.75, rng=rng))
t0 = time.perf_counter()
hist = []
for i in range(N):
    lr, b1 = sched.at(i)
    opt.set_hyper(lr=lr, beta1=b1)
    l = step(None, draw_block_mask((40, 40, 40), 0.75, rng=rng))
    if i % 10 == 0 or i == N - 1:
        v = [x.item() for x in l]
        hist.append(v[0])
        print(f"step {i:4d}  loss {v[0]:.4f}  rgb {v[1]:.4f}  alpha {v[2]:.4f}  |g| {opt.norm.item():.3f}", flush=True)
torch.cuda.synchronize()
print(f"{N} steps in {time.perf_counter() - t0:.2f} s; finite: {all(h == h for h in hist)}; first {hist[0]:.4f} -> last {hist[-1]:.4f}")
