"""Input pipeline: oracle restatement vs the reference's own functions (golden g11), host logic, and the
HIP kernel vs the oracle."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import mae3d_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden", "g11_input_pipeline.npz")
R = 12


def _scene(seed, shape, u8=False):   # same generator as oracle/gen_golden_input.py
    rng = np.random.default_rng(seed)
    g = rng.standard_normal(tuple(shape) + (4,)).astype(np.float32) * 2.0
    if u8:
        return rng.integers(0, 256, tuple(shape) + (4,), dtype=np.uint8)
    return g


def _cases():
    g = np.load(GOLD)
    for s, w, l, h, u8 in g["cases"]:
        for aug in range(6):
            yield g, int(s), (int(w), int(l), int(h)), bool(u8), aug


def test_oracle_matches_reference_golden():
    n = 0
    for g, s, sh, u8, aug in _cases():
        random.seed(100 * s + aug)
        flags = O.draw_augmentation(0.5, 0.5, random) | (0 if u8 else O.GRID_DENSITY)
        out, ext = O.prepare_grid(_scene(s, sh, u8), R, flags)
        np.testing.assert_array_equal(np.array(ext), g[f"c{s}_a{aug}_ext"])
        np.testing.assert_allclose(out.numpy(), g[f"c{s}_a{aug}"], rtol=1e-6, atol=1e-7)
        assert abs(float(ext[0] * ext[1] * ext[2] * 4) - float(g[f"c{s}_a{aug}_masksum"][0])) < 0.5   # the ones-mask == the extents
        n += 1
    assert n == 24


def test_host_draw_order_matches_oracle():
    from nerf_mae_amd import data, ops
    assert (ops.GRID_ROT, ops.GRID_FLIP0, ops.GRID_FLIP1, ops.GRID_DENSITY) == (O.GRID_ROT, O.GRID_FLIP0, O.GRID_FLIP1, O.GRID_DENSITY)
    for seed in range(50):
        a, b = random.Random(seed), random.Random(seed)
        assert data.draw_augmentation(0.5, 0.3, a) == O.draw_augmentation(0.5, 0.3, b)
    with pytest.raises(ValueError):
        data.draw_augmentation(1.5, 0.0)
    sc = data.synthetic_scene((10, 8, 6), 3)
    assert sc.shape == (10, 8, 6, 4) and sc.dtype == np.float32
    assert data.synthetic_scene((10, 8, 6), 3, dtype=np.uint8).dtype == np.uint8


@pytest.mark.gpu
def test_grid_prepare_kernel_matches_golden_and_oracle():
    from nerf_mae_amd import data, ops
    for g, s, sh, u8, aug in _cases():
        random.seed(100 * s + aug)
        flags = O.draw_augmentation(0.5, 0.5, random) | (0 if u8 else O.GRID_DENSITY)
        src = torch.from_numpy(_scene(s, sh, u8)).cuda()
        dst = torch.full((4, R, R, R), 7.0, device="cuda")
        ext = ops.grid_prepare(src, dst, R, flags)
        assert tuple(ext) == tuple(int(v) for v in g[f"c{s}_a{aug}_ext"])
        np.testing.assert_allclose(dst.cpu().numpy(), g[f"c{s}_a{aug}"], rtol=2e-6, atol=2e-7)
    # the batcher: staging through pinned memory, per-sample flags, extents tensor; full-size scene incl. uint8
    bt = data.GridBatcher(160, "cuda", normalize_density=True)
    scenes = [data.synthetic_scene((160, 132, 96), 1), data.synthetic_scene((120, 160, 144), 2, dtype=np.uint8)]
    flags = [ops.GRID_ROT | ops.GRID_FLIP1, ops.GRID_FLIP0]
    xb, ext = bt(scenes, flags=flags)
    assert ext.cpu().tolist() == [[132, 160, 96], [120, 160, 144]]
    for i, sc in enumerate(scenes):
        ref, _ = O.prepare_grid(sc, 160, flags[i] | (O.GRID_DENSITY if sc.dtype != np.uint8 else 0))
        np.testing.assert_allclose(xb[i].cpu().numpy(), ref.numpy(), rtol=2e-6, atol=2e-7)
    with pytest.raises(Exception):
        ops.grid_prepare(torch.zeros(200, 4, 4, 4, device="cuda"), xb[0], 160, 0)   # does not fit the resolution


@pytest.mark.gpu
def test_staging_ring_survives_host_running_ahead():
    """the host queues six batches behind a device that is still busy (a long sleep kernel on the stream): every batch must arrive
    intact.  With one pinned buffer per slot (round 1) the host overwrote the buffer of a copy that had not executed yet; the ring's
    per-buffer events make the host wait instead.  Contents are verified on the device (exact fp32 sums of uint8-derived grids)."""
    import numpy as np
    from nerf_mae_amd import data
    R = 32
    bt = data.GridBatcher(R, "cuda", normalize_density=True, depth=3)
    scenes = [[data.synthetic_scene((32, 30 - b, 28), seed=100 * b + i, dtype=np.uint8) for i in range(2)] for b in range(6)]
    torch.cuda._sleep(400_000_000)   # ~0.2 s of device time: all six calls below are issued while it runs
    outs = [bt.prepare(batch, flags=[0, 0]) for batch in scenes]
    torch.cuda.synchronize()
    for (xb, ext), batch in zip(outs, scenes):
        for i, sc in enumerate(batch):
            want = torch.from_numpy(sc.astype(np.float32) / 255.0).permute(3, 0, 1, 2)
            a0, a1, a2 = ext[i]
            assert (a0, a1, a2) == tuple(sc.shape[:3])
            got = xb[i, :, :a0, :a1, :a2].cpu()
            assert torch.allclose(got, want, rtol=2e-6, atol=1e-7), (i, (got - want).abs().max())
            assert abs(float(xb[i].double().sum()) - float(got.double().sum())) < 1e-9   # the padding is zero


@pytest.mark.gpu
def test_prefetcher_delivers_batches_in_order_with_a_slow_and_a_fast_consumer():
    """background-thread pipeline (pinned rings -> copy stream -> grid_prepare): every batch equals the synchronous batcher's output,
    also when the consumer is slower than the producer (device buffers are only reused after `done`) and for a ragged last batch"""
    import time
    import numpy as np
    from nerf_mae_amd import data
    R = 32
    scenes = [data.synthetic_scene((32, 32 - (i % 3), 29), seed=i, dtype=np.float32 if i % 2 else np.uint8) for i in range(7)]
    batches = [scenes[0:2], scenes[2:4], scenes[4:6], scenes[6:7]]
    ref = data.GridBatcher(R, "cuda", normalize_density=True)
    for delay in (0.0, 0.05):
        pf = data.Prefetcher(data.GridBatcher(R, "cuda", normalize_density=True), batches, 2)
        main = torch.cuda.current_stream()
        n = 0
        for j, xb, ext, ev in pf:
            main.wait_event(ev)
            got = xb.clone()
            pf.done(j, main)
            want, wext = ref.prepare(batches[n], flags=[0] * len(batches[n]))
            torch.cuda.synchronize()
            assert ext == wext and torch.equal(got, want), n
            time.sleep(delay)
            n += 1
        assert n == len(batches)


@pytest.mark.gpu
def test_prefetcher_never_overwrites_a_buffer_before_done_with_a_device_backlog():
    """the consumer's stream has a long backlog in front of its device-to-device copy (the previous step's graph replay) and it calls
    `done` late: a buffer must stay untouched from hand-out until the event `done` records has completed.  (Round 2's prefetcher let
    the producer proceed when `done` had not been called yet -- it saw `None` or the stale event of batch n-4 -- and `grid_prepare` on
    the high-priority copy stream then overwrote the batch the training step was about to read.)"""
    import time
    import numpy as np
    from nerf_mae_amd import data
    R = 32
    scenes = [data.synthetic_scene((32, 31 - (i % 2), 30), seed=50 + i, dtype=np.uint8) for i in range(12)]
    batches = [scenes[2 * b:2 * b + 2] for b in range(6)]
    ref = data.GridBatcher(R, "cuda", normalize_density=True)
    want = [ref.prepare(b, flags=[0, 0])[0].clone() for b in batches]
    torch.cuda.synchronize()
    bt = data.GridBatcher(R, "cuda", normalize_density=True)
    pf = data.Prefetcher(bt, batches, 2, depth=2)
    main = torch.cuda.current_stream()
    got = []
    for j, xb, ext, ev in pf:
        main.wait_event(ev)
        torch.cuda._sleep(60_000_000)      # ~30 ms of device time queued in front of the read of this batch
        got.append(xb.clone())             # the read (queued behind the sleep)
        time.sleep(0.03)                   # the window in which the old producer started the next-but-one batch into the same buffer
        pf.done(j, main)
    torch.cuda.synchronize()
    assert len(got) == len(want)
    for n, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), f"batch {n} was overwritten before its read executed"


@pytest.mark.gpu
def test_prefetcher_thread_ends_when_the_consumer_leaves_early():
    """a consumer that stops iterating (exception in the step, `break`) without handing its buffers back must not leave the producer thread spinning on
    them: leaving the iterator closes the prefetcher, and the thread is gone shortly after"""
    import time
    import numpy as np
    from nerf_mae_amd import data
    R = 32
    scenes = [data.synthetic_scene((32, 32, 30), seed=70 + i, dtype=np.uint8) for i in range(16)]
    batches = [scenes[2 * b:2 * b + 2] for b in range(8)]
    pf = data.Prefetcher(data.GridBatcher(R, "cuda", normalize_density=True), batches, 2, depth=2)
    for j, xb, ext, ev in pf:
        torch.cuda.current_stream().wait_event(ev)
        break                                  # no done(j): the buffer is never handed back
    t0 = time.time()
    while pf.t.is_alive() and time.time() - t0 < 5.0:
        time.sleep(0.05)
    assert not pf.t.is_alive(), "the producer thread survived its consumer"
    pf.close()                                 # idempotent


def test_prefetcher_uses_a_private_rng():
    """augmentation flags come from the prefetcher's own generator: drawing them leaves the global `random` stream (the training
    thread's block masks) untouched, and equal seeds give equal flags whatever else runs"""
    import random
    from nerf_mae_amd import data

    class _B:   # batcher stand-in: records the flags, no device
        flip_prob, rotate_prob, R, device = 0.5, 0.5, 8, "cpu"

    def flags_for(seed):
        rng = random.Random(seed)
        return [data.draw_augmentation(0.5, 0.5, rng) for _ in range(16)]
    random.seed(5)
    before = random.getstate()
    a, b = flags_for(3), flags_for(3)
    assert a == b and random.getstate() == before and flags_for(4) != a
    import inspect
    sig = inspect.signature(data.Prefetcher.__init__)
    assert sig.parameters["rng"].default is None and "seed" in sig.parameters
