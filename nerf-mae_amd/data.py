"""Input pipeline of the MAE pretraining path: stored scenes -> padded device batch.

The reference loads `.npz['rgbsigma']` (W,L,H,4) on the host, applies density->alpha, transposes to (4,W,L,H), augments with
90-degree rotations / flips and pads to (4,R,R,R) in PyTorch on the CPU, then ships 65.5 MB of fp32 (plus an equally large
ones-mask) per grid to the GPU (nerf_rpn/datasets.py:88-101,198-233,247-248; torch_utils.py:56-90).  Here the scene crosses PCIe as
stored (uint8 scenes: 16 MB instead of 131 MB) through a pinned staging buffer and ONE HIP kernel (`nmh_grid_prepare`) does
normalisation, density->alpha, layout change, augmentation and zero padding straight into the batch tensor; the ones-mask is
replaced by three integers per sample (the valid extents).  No arithmetic of the path runs on the CPU.
"""
import random as _random
from typing import List, Optional, Sequence, Tuple, Union

import os

import numpy as np
import torch

from . import ops

Scene = Union[np.ndarray, torch.Tensor]


def load_scene(path: str) -> np.ndarray:
    """`rgbsigma` of one stored scene, (W,L,H,4) float32 or uint8, untouched (data/scannet/run_nerf.py:1904-1913 schema)."""
    with np.load(path) as f:
        return np.ascontiguousarray(f["rgbsigma"])


def draw_augmentation(flip_prob: float, rotate_prob: float, rng=_random) -> int:
    """flags for `ops.grid_prepare`, drawn in the reference's order (datasets.py:198-233 with boxes=None, z_up=True):
    one draw for the rotation, then one per flip axis"""
    if not (0.0 <= flip_prob <= 1.0):
        raise ValueError("flip_prob must be between 0 and 1, but got {}".format(flip_prob))
    if not (0.0 <= rotate_prob <= 1.0):
        raise ValueError("rotate_prob must be between 0 and 1, but got {}".format(rotate_prob))
    flags = 0
    if rng.random() < rotate_prob:
        flags |= ops.GRID_ROT
    if rng.random() < flip_prob:
        flags |= ops.GRID_FLIP0
    if rng.random() < flip_prob:
        flags |= ops.GRID_FLIP1
    return flags


def synthetic_scene(shape: Sequence[int] = (160, 160, 160), seed: int = 0, dtype=np.float32) -> np.ndarray:
    """A stored-format synthetic scene (W,L,H,4) following SURVEY 8(d): RGB ~ U[0,1); raw density sigma ~ N(0,3^2) inside the
    central 60 % box and -10 outside (float32), so that density->alpha gives a realistic occupied fraction; uint8: RGB and an
    already-normalised alpha, scaled to 0..255."""
    rng = np.random.default_rng(seed)
    W, L, H = shape
    g = np.empty((W, L, H, 4), dtype=np.float32)
    g[..., :3] = rng.random((W, L, H, 3), dtype=np.float32)
    sigma = np.full((W, L, H), -10.0, dtype=np.float32)
    lo, hi = [int(0.2 * s) for s in shape], [int(0.8 * s) for s in shape]
    sigma[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = rng.standard_normal((hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]), dtype=np.float32) * 3.0
    if dtype == np.uint8:
        alpha = np.clip(1.0 - np.exp(-np.exp(sigma) / 100.0), 0.0, 1.0)
        g[..., 3] = alpha
        return np.round(g * 255.0).astype(np.uint8)
    g[..., 3] = sigma
    return g


def _wait_event(ev):
    """host-side wait by polling: a blocking hipEventSynchronize from the producer thread was measured to stall the kernel / graph
    launches of the training thread for the whole wait (runtime-internal locking), tripling the step time"""
    import time
    while not ev.query():
        time.sleep(2e-4)


class _PinnedSlot:
    """ring of pinned host buffers + device staging buffers for one slot of the batch; a buffer is rewritten only after the event
    recorded behind its previous H2D copy + grid_prepare launch has completed (the host may run several steps ahead of the device)"""

    def __init__(self, depth: int, device):
        self.depth, self.device = depth, device
        self.pin, self.dev, self.ev, self.i = [None] * depth, [None] * depth, [None] * depth, 0

    def acquire(self, numel: int, dtype):
        k = self.i % self.depth
        self.i += 1
        if self.ev[k] is not None:
            _wait_event(self.ev[k])
            self.ev[k] = None
        if self.pin[k] is None or self.pin[k].numel() < numel or self.pin[k].dtype != dtype:
            self.pin[k] = torch.empty(numel, dtype=dtype).pin_memory()
            self.dev[k] = torch.empty(numel, dtype=dtype, device=self.device)
        return k, self.pin[k][:numel], self.dev[k][:numel]

    def release_after(self, k: int, stream):
        ev = torch.cuda.Event()
        ev.record(stream)
        self.ev[k] = ev


class GridBatcher:
    """Turns a list of stored scenes into the network input `(xb (B,4,R,R,R) fp32, extents (B,3) int32)` on `device`.

    Host side: a ring of `depth` pinned staging buffers per slot of the batch, each guarded by an event (see _PinnedSlot), so a
    caller that runs ahead of the device never overwrites a buffer whose H2D copy has not executed yet.  Device side:
    `ops.grid_prepare` per scene on the current stream.  `normalize_density` applies to float scenes (uint8 scenes are stored with
    alpha already normalised, as the reference's /255 branch assumes).  `Prefetcher` runs this on a background thread + copy stream."""

    def __init__(self, resolution: int, device, normalize_density: bool = True, flip_prob: float = 0.0, rotate_prob: float = 0.0, depth: int = 3):
        self.R, self.device = resolution, torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("GridBatcher needs a HIP device (no CPU fallback)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.normalize_density, self.flip_prob, self.rotate_prob, self.depth = normalize_density, flip_prob, rotate_prob, depth
        self._slots = {}

    _pool = None

    @classmethod
    def _copy_pool(cls):
        # host -> pinned copies of one batch run on a few worker threads (the copy releases the GIL; one thread moves a 66 MB fp32 scene
        # at 5-7 GB/s: 36 ms per batch of 4, more than the 33 ms training step it has to hide under)
        if cls._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            cls._pool = ThreadPoolExecutor(max_workers=int(os.environ.get("NMH_STAGE_THREADS", "8")), thread_name_prefix="nmh-stage")
        return cls._pool

    def _stage_begin(self, slot: int, scene: Scene):
        """-> (host tensor, ring index, pinned buffer, device buffer, host-copy futures); a scene that is already on the device comes back as
        (scene, None, None, None, [])"""
        if isinstance(scene, torch.Tensor) and scene.is_cuda:
            return scene.contiguous(), None, None, None, []
        t = torch.from_numpy(scene) if isinstance(scene, np.ndarray) else scene
        t = t.contiguous()
        sl = self._slots.get(slot)
        if sl is None:
            sl = self._slots[slot] = _PinnedSlot(self.depth, self.device)
        k, pin, dev = sl.acquire(t.numel(), t.dtype)
        futs = []
        if not t.is_pinned():
            flat, n = t.reshape(-1), t.numel()
            nchunk = max(1, min(4, n * t.element_size() // (8 << 20)))     # >= 8 MB per task
            step = (n + nchunk - 1) // nchunk
            pool = self._copy_pool()
            for a in range(0, n, step):
                futs.append(pool.submit(pin[a:a + step].copy_, flat[a:a + step]))
        return t, k, pin, dev, futs

    @staticmethod
    def _stage_end(t, k, pin, dev, futs):
        if k is None:
            return t
        for f in futs:
            f.result()
        # (already page-locked scenes -- data.pin_scene -- are read in place: no host-side copy)
        dev.copy_(t.reshape(-1) if t.is_pinned() else pin, non_blocking=True)
        return dev.view(t.shape)

    def _stage(self, slot: int, scene: Scene):
        """-> (device tensor holding the stored scene, ring index or None)"""
        st = self._stage_begin(slot, scene)
        return self._stage_end(*st), st[1]

    def __call__(self, scenes: List[Scene], flags: Optional[List[int]] = None, out: Optional[torch.Tensor] = None,
                 rng=_random) -> Tuple[torch.Tensor, torch.Tensor]:
        xb, ext = self.prepare(scenes, flags, out, rng)
        return xb, torch.tensor(ext, dtype=torch.int32).to(self.device, non_blocking=True)

    def prepare(self, scenes: List[Scene], flags: Optional[List[int]] = None, out: Optional[torch.Tensor] = None, rng=_random):
        """as __call__, but the extents come back as a host list (no device round trip)"""
        B, R = len(scenes), self.R
        xb = out if out is not None else torch.empty((B, 4, R, R, R), dtype=torch.float32, device=self.device)
        ext = []
        st = torch.cuda.current_stream()
        staged = [self._stage_begin(i, sc) for i, sc in enumerate(scenes)]     # all host copies of the batch in flight at once
        for i, sc in enumerate(scenes):
            f = flags[i] if flags is not None else draw_augmentation(self.flip_prob, self.rotate_prob, rng)
            src, k = self._stage_end(*staged[i]), staged[i][1]
            if self.normalize_density and src.dtype == torch.float32:
                f |= ops.GRID_DENSITY
            ext.append(list(ops.grid_prepare(src, xb[i], R, f)))
            if k is not None:
                self._slots[i].release_after(k, st)
        return xb, ext


def pin_scene(scene: np.ndarray) -> torch.Tensor:
    """page-locked copy of a stored scene: datasets that fit in host memory can be pinned once, after which every epoch's H2D copy
    reads the scene in place"""
    return torch.from_numpy(np.ascontiguousarray(scene)).pin_memory()


class Prefetcher:
    """Asynchronous input pipeline (the reference's DataLoader(num_workers=2, pin_memory=True), run_swin_mae3d.py:578-586): a
    background thread loads the scenes of batch k+1, stages them through the pinned rings and runs the H2D copies and the
    `grid_prepare` kernels on its own copy stream while step k computes; the consumer gets `(xb, extents, event)` and makes its
    stream wait for the event.  `depth` device batches rotate.  A buffer that has been handed out is PENDING until the consumer
    calls `done(slot)` -- which records the event after which it may be overwritten -- and ONLY `done` frees it: the producer first
    waits for the hand-back, then for that event (a consumer whose stream still has the device-to-device copy of batch n-2 queued
    behind a long replay can therefore never have buffer n%depth overwritten underneath it, whatever the thread timing).  The
    consumer MUST call `done` once per delivered batch.  Augmentation flags come from the prefetcher's OWN `random.Random`
    (`seed`), never from the global module the training thread draws its block masks from: both streams stay reproducible."""

    _PENDING = object()

    def __init__(self, batcher: GridBatcher, batches, batch_size: int, load=None, depth: int = 2, rng=None, seed: int = 0):
        import queue
        import threading
        # rng=None: a private generator (the global `random` module belongs to the training thread's mask draws)
        self.b, self.load, self.rng = batcher, load or (lambda s: s), (rng if rng is not None else _random.Random(seed))
        R, dev = batcher.R, batcher.device
        self.bufs = [torch.empty((batch_size, 4, R, R, R), dtype=torch.float32, device=dev) for _ in range(depth)]
        self.free = [None] * depth                      # None: never handed out; _PENDING: with the consumer; else the event after which buffer j may be overwritten
        self._handback = threading.Condition()
        self.stream = torch.cuda.Stream(device=dev, priority=-1)   # high priority: its small copy/prepare kernels slot in between the step's kernels
        self.q = queue.Queue(maxsize=depth - 1 if depth > 1 else 1)
        self.batches = batches
        self.err = None
        self._stop = False                              # close(): the producer leaves its wait loops and exits
        # the producer re-acquires the GIL after every C call (host copy, H2D, kernel launch: ~25 per batch); with CPython's default 5 ms
        # switch interval each hand-over from a busy consumer thread can take that long (measured: a 66 MB host copy 0.07 ms alone, 11 ms
        # next to a spinning Python thread) -- 0.2 ms keeps the producer's latency per call small against a 13-60 ms step
        import sys
        if sys.getswitchinterval() > 2e-4:
            sys.setswitchinterval(2e-4)
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _run(self):
        try:
            torch.cuda.set_device(self.b.device)
            import time
            st = self.stats = {"batches": 0, "load_s": 0.0, "wait_free_s": 0.0, "prepare_s": 0.0, "put_s": 0.0}   # producer-side wall clock per phase
            for n, batch in enumerate(self.batches):
                j = n % len(self.bufs)
                t0 = time.perf_counter()
                scenes = [self.load(s) for s in batch]
                flags = [draw_augmentation(self.b.flip_prob, self.b.rotate_prob, self.rng) for _ in scenes]
                t1 = time.perf_counter()
                with self._handback:                     # handed out before: wait until the consumer has given it back (done(j)) ...
                    while self.free[j] is Prefetcher._PENDING and not self._stop:
                        self._handback.wait(0.05)
                if self._stop:
                    return
                if self.free[j] is not None:
                    _wait_event(self.free[j])           # ... and until its last read of the buffer has executed
                t2 = time.perf_counter()
                with torch.cuda.stream(self.stream):
                    xb, ext = self.b.prepare(scenes, flags, out=self.bufs[j][:len(scenes)])
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                t3 = time.perf_counter()
                self.free[j] = Prefetcher._PENDING
                if not self._put((j, xb, ext, ev)):
                    return
                t4 = time.perf_counter()
                st["batches"] += 1; st["load_s"] += t1 - t0; st["wait_free_s"] += t2 - t1; st["prepare_s"] += t3 - t2; st["put_s"] += t4 - t3
            self._put(None)
        except BaseException as e:  # noqa: BLE001
            self.err = e
            self._put(None)

    def _put(self, item) -> bool:
        """bounded put: a consumer that went away must not park the thread for good (also for the terminating None).  False: stopped, `item` was dropped"""
        import queue
        while not self._stop:
            try:
                self.q.put(item, timeout=0.05)
                return True
            except queue.Full:
                continue
        return False

    def __iter__(self):
        try:
            while True:
                item = self.q.get()
                if item is None:
                    if self.err is not None:
                        raise self.err
                    return
                yield item
        finally:
            self.close()                                 # a consumer that leaves early (exception, break) releases the producer thread

    def close(self):
        """stop the producer (idempotent): it leaves its wait loops within 50 ms and drops its references to the staged batches"""
        import queue
        import threading
        self._stop = True
        with self._handback:
            self._handback.notify_all()

        def drain():
            try:
                while True:
                    self.q.get_nowait()
            except queue.Empty:
                pass
        drain()
        # the producer may have passed its `_stop` check just before the drain and still put one more staged batch: wait for it to exit (it leaves every
        # wait loop within 50 ms), then drain again so that no batch -- and no device buffer -- stays referenced by the queue
        t = getattr(self, "t", None)
        if t is not None and t.is_alive() and t is not threading.current_thread():
            t.join(timeout=1.0)
        drain()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def done(self, j: int, stream=None):
        """the consumer has queued its last read of buffer j on `stream`"""
        ev = torch.cuda.Event()
        ev.record(stream or torch.cuda.current_stream())
        with self._handback:
            self.free[j] = ev
            self._handback.notify_all()
