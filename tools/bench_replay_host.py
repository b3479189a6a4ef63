"""host time of one GraphedTrainStep call (graph replay + step_params launch, no synchronisation) against the device time of the step"""
import sys, time, random, torch
sys.path.insert(0, '.')
from nerf_mae_amd import ops
from nerf_mae_amd.model import build_model, draw_block_mask
from nerf_mae_amd.trainer import FusedAdamW, GraphedTrainStep
from oracle import mae3d_oracle as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ops.side_stream.auto(B)
m = build_model('swin_s', 160, 0.75, 0.1).cuda().train(); m.flatten_parameters()
opt = FusedAdamW(m, lr=1e-4, weight_decay=1e-3, max_grad_norm=0.1)
grids = [O.synthetic_grid((160, 160, 160), i).cuda() for i in range(B)]
rng = random.Random(0)
step = GraphedTrainStep(m, opt, B)
step(grids, draw_block_mask((40, 40, 40), 0.75, rng=rng))
for _ in range(3): step(None, draw_block_mask((40, 40, 40), 0.75, rng=rng))
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N): step(None, draw_block_mask((40, 40, 40), 0.75, rng=rng))
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host {1e3 * (t1 - t0) / N:.2f} ms per call (queued without waiting), device-bound total {1e3 * (t2 - t0) / N:.2f} ms per step")
