"""per-step wall time of the first replays after capture (8 grids per GPU): does a short --steps/--warmup run see a slower start?"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_mae_amd import data
from nerf_mae_amd.model import build_model, draw_block_mask
from nerf_mae_amd.trainer import FusedAdamW, GraphedTrainStep, OneCycle
dev = torch.device("cuda", 0)
R, g, nb = 160, 40, int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0); random.seed(0)
model = build_model("swin_s", resolution=R, masking_prob=0.75, stochastic_depth_prob=0.1, compute_dtype=torch.bfloat16).to(dev)
model.train(); model.flatten_parameters()
opt = FusedAdamW(model, lr=1e-4, weight_decay=1e-3, max_grad_norm=0.1)
sched = OneCycle(1e-4, 1000); rng = random.Random(1)
scenes = [data.synthetic_scene((R, R, R), seed=i) for i in range(nb)]
xb, ext = data.GridBatcher(R, dev, normalize_density=True)(scenes, flags=[0] * nb)
grids = [xb[i].contiguous() for i in range(nb)]
gs = GraphedTrainStep(model, opt, nb)
gs(grids, draw_block_mask((g, g, g), 0.75, rng=rng))
ts = []
for n in range(30):
    lr, b1 = sched.at(n); opt.set_hyper(lr=lr, beta1=b1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gs(None, draw_block_mask((g, g, g), 0.75, rng=rng))
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("per-step ms (synchronised each step):", " ".join(f"{t:.2f}" for t in ts))
