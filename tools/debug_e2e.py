import sys, time, random
import numpy as np, torch
sys.path.insert(0, '.')
from nerf_mae_amd import data
from nerf_mae_amd.model import build_model, draw_block_mask
from nerf_mae_amd.trainer import Trainer
R = 160
model = build_model("swin_s", resolution=R).cuda(); model.train(); model.flatten_parameters()
scenes = [data.synthetic_scene((R, R, R), seed=50 + i, dtype=np.uint8) for i in range(8)]
tr = Trainer(model, scenes * 5, batch_size=4, num_epochs=3, log=lambda *_: None)
tr.train_epoch(1); torch.cuda.synchronize()
# instrumented epoch
orig_prepare = tr.batcher.prepare
pt = []
def timed_prepare(*a, **k):
    t0 = time.perf_counter(); r = orig_prepare(*a, **k); pt.append(time.perf_counter() - t0); return r
tr.batcher.prepare = timed_prepare
t0 = time.perf_counter(); tr.train_epoch(2); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("epoch: %.1f ms/step; producer prepare() host times (ms):" % (1e3 * dt / tr.steps_per_epoch), [round(1e3 * t, 1) for t in pt])
# consumer-side breakdown
from nerf_mae_amd import data as D
idx = list(range(40)); batches = [[scenes[i % 8] for i in idx[b*4:(b+1)*4]] for b in range(10)]
pf = D.Prefetcher(tr.batcher, batches, 4)
main = torch.cuda.current_stream(); g = R // 4
tw = []; ts = []
t_prev = time.perf_counter()
for j, xb, ext, ev in pf:
    t1 = time.perf_counter(); tw.append(t1 - t_prev)
    main.wait_event(ev); ta = time.perf_counter(); tr.step_fn.x[:4].copy_(xb, non_blocking=True); tb = time.perf_counter(); pf.done(j, main); tr.step_fn.set_extents(ext)
    tr.opt.set_hyper(lr=1e-5, beta1=0.9)
    bm = draw_block_mask((g, g, g), 0.75, rng=random)
    tc = time.perf_counter()
    td = time.perf_counter()
    tr.step_fn(None, bm); te = tf = time.perf_counter()
    t_prev = time.perf_counter(); ts.append([round(1e3 * v, 1) for v in (ta - t1, tb - ta, tc - tb, td - tc, te - td, tf - te)])
torch.cuda.synchronize()
print("consumer wait for batch (ms):", [round(1e3 * t, 1) for t in tw]); print("consumer [wait_event, d2d copy, done+ext+mask draw, mask upload, hyper upload, replay] (ms):"); [print("  ", t) for t in ts]
