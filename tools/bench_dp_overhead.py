"""What the data-parallel machinery costs on one GPU: the captured training step at G grids per GPU with a GradReducer over a ONE-rank RCCL
group (split graphs, bf16 bucket casts, comm-stream events; the all-reduce itself degenerates to a copy) against the plain single-process
step.  An upper bound on the per-GPU compute side of an N-GPU run; the exposed part of the real all-reduce comes on top.
usage: python tools/bench_dp_overhead.py [grids per GPU ...]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from nerf_mae_amd import data
from nerf_mae_amd.dist import GradReducer, broadcast_parameters
from nerf_mae_amd.model import build_model, draw_block_mask
from nerf_mae_amd.trainer import FusedAdamW, GraphedTrainStep, OneCycle

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
R, g = 160, 40
torch.manual_seed(0); random.seed(0)
model = build_model("swin_s", resolution=R, masking_prob=0.75, stochastic_depth_prob=0.1, compute_dtype=torch.bfloat16).to(dev)
model.train(); model.flatten_parameters(); broadcast_parameters(model)
opt = FusedAdamW(model, lr=1e-4, weight_decay=1e-3, max_grad_norm=0.1)
sched = OneCycle(1e-4, 1000)
rng = random.Random(1)
# modes over a ONE-rank RCCL group: the collectives degenerate to copies, everything around them is real
MODES = (("single process (one graph, no exchange)", {}, False),
         ("split graphs, collectives between the replays", {"NMH_DP_FORCE_SPLIT": "1", "NMH_DP_FORCE": "1"}, True),
         ("ONE graph with the RCCL all-reduces captured inside", {"NMH_DP_FORCE": "1", "NMH_DP_CAPTURE_COMM": "1"}, True))
for nb in [int(a) for a in sys.argv[1:]] or [1, 2]:
    scenes = [data.synthetic_scene((R, R, R), seed=i) for i in range(nb)]
    xb, ext = data.GridBatcher(R, dev, normalize_density=True)(scenes, flags=[0] * nb)
    grids = [xb[i].contiguous() for i in range(nb)]
    for name, env, use_red in MODES:
        for k in ("NMH_DP_FORCE_SPLIT", "NMH_DP_FORCE", "NMH_DP_CAPTURE_COMM"):
            os.environ.pop(k, None)
        os.environ.update(env)
        red = GradReducer(model, comm_dtype=torch.bfloat16) if use_red else None
        model._reducer = None
        gs = GraphedTrainStep(model, opt, nb, reducer=red)
        gs(grids, draw_block_mask((g, g, g), 0.75, rng=rng))
        n = 0
        def step():
            global n
            lr, b1 = sched.at(n); n += 1
            opt.set_hyper(lr=lr, beta1=b1)
            return gs(None, draw_block_mask((g, g, g), 0.75, rng=rng))[0]
        for _ in range(5): loss = step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): loss = step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"{nb} grid(s)/GPU, {name:52s}: {dt * 1e3:7.3f} ms/step   loss {loss.item():.4f}   comm_captured={gs.comm_captured}", flush=True)
        del gs, red
        model._reducer = None
        torch.cuda.empty_cache()
dist.destroy_process_group()
