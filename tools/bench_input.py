"""input pipeline micro-benchmark: host memcpy into pinned memory, H2D, GridBatcher.prepare, Prefetcher alone (no training)"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from nerf_mae_amd import data
R = 160
for dt in (np.uint8, np.float32):
    scenes = [data.synthetic_scene((R, R, R), seed=i, dtype=dt) for i in range(8)]
    t = torch.from_numpy(scenes[0]).reshape(-1)
    pin = torch.empty(t.numel(), dtype=t.dtype).pin_memory()
    dev = torch.empty(t.numel(), dtype=t.dtype, device="cuda")
    pin.copy_(t); dev.copy_(pin); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): pin.copy_(t)
    t1 = time.perf_counter()
    for _ in range(5): dev.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    mb = t.numel() * t.element_size() / 1e6
    print(f"{dt.__name__}: scene {mb:.0f} MB; host->pinned {mb * 5 / (t1 - t0) / 1e3:.1f} GB/s; H2D {mb * 5 / (t2 - t1) / 1e3:.1f} GB/s")
    bt = data.GridBatcher(R, "cuda")
    for rep in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        xb, ext = bt.prepare(scenes[:4], flags=[0] * 4)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"  prepare(4 scenes): host {1e3 * (t1 - t0):.1f} ms, to completion {1e3 * (t2 - t0):.1f} ms")
    batches = [scenes[0:4], scenes[4:8]] * 5
    pf = data.Prefetcher(bt, batches, 4)
    main = torch.cuda.current_stream()
    t0 = time.perf_counter()
    n = 0
    for j, xb, ext, ev in pf:
        main.wait_event(ev)
        pf.done(j, main)
        n += 1
    torch.cuda.synchronize()
    print(f"  Prefetcher alone: {1e3 * (time.perf_counter() - t0) / n:.1f} ms per batch of 4 -> {4 * n / (time.perf_counter() - t0):.0f} grids/s")
