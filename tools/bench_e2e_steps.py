"""Trainer.fit from host scenes: ms/step as a function of the steps per epoch (pipeline fill + thread start-up amortisation)"""
import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from nerf_mae_amd import data, ops
from nerf_mae_amd.model import build_model
from nerf_mae_amd.trainer import Trainer
R, nb = 160, 4
ops.side_stream.auto(nb)
model = build_model('swin_s', R, 0.75, 0.1).cuda()
import sys as _s
DT = np.float32 if (len(_s.argv) > 1 and _s.argv[1] == 'fp32') else np.uint8
scenes = [data.synthetic_scene((R, R, R), seed=50 + i, dtype=DT) for i in range(8)]
for mult in (32,):
    tr = Trainer(model, scenes * mult, batch_size=nb, num_epochs=1, log=lambda *_: None)
    tr.train_epoch(1)
    for ep in (2, 3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.train_epoch(ep)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ps = tr.prefetch_stats or {}
        nb_ = max(1, ps.get("batches", 1))
        print(f"steps/epoch {tr.steps_per_epoch}: {1e3 * dt / tr.steps_per_epoch:.2f} ms/step; producer per batch (ms):", {k: round(1e3 * v / nb_, 2) for k, v in ps.items() if k != "batches"})
    del tr
