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


class GridBatcher:
    """Turns a list of stored scenes into the network input `(xb (B,4,R,R,R) fp32, extents (B,3) int32)` on `device`.

    Host side: one pinned staging buffer per slot of the batch (reused every step; the copy engine overlaps the previous step's
    compute when the caller runs on a side stream).  Device side: `ops.grid_prepare` per scene.  `normalize_density` applies to
    float scenes (uint8 scenes are stored with alpha already normalised, as the reference's /255 branch assumes)."""

    def __init__(self, resolution: int, device, normalize_density: bool = True, flip_prob: float = 0.0, rotate_prob: float = 0.0):
        self.R, self.device = resolution, torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("GridBatcher needs a HIP device (no CPU fallback)")
        self.normalize_density, self.flip_prob, self.rotate_prob = normalize_density, flip_prob, rotate_prob
        self._pinned = {}
        self._staged = {}

    def _stage(self, slot: int, scene: Scene) -> torch.Tensor:
        if isinstance(scene, torch.Tensor) and scene.is_cuda:
            return scene.contiguous()
        t = torch.from_numpy(scene) if isinstance(scene, np.ndarray) else scene
        t = t.contiguous()
        key = (slot, t.dtype)
        pin = self._pinned.get(key)
        if pin is None or pin.numel() < t.numel():
            pin = self._pinned[key] = torch.empty(t.numel(), dtype=t.dtype).pin_memory()
            self._staged[key] = torch.empty(t.numel(), dtype=t.dtype, device=self.device)
        pin[: t.numel()].copy_(t.reshape(-1))
        dev = self._staged[key][: t.numel()]
        dev.copy_(pin[: t.numel()], non_blocking=True)
        return dev.view(t.shape)

    def __call__(self, scenes: List[Scene], flags: Optional[List[int]] = None, out: Optional[torch.Tensor] = None,
                 rng=_random) -> Tuple[torch.Tensor, torch.Tensor]:
        B, R = len(scenes), self.R
        xb = out if out is not None else torch.empty((B, 4, R, R, R), dtype=torch.float32, device=self.device)
        ext = []
        for i, sc in enumerate(scenes):
            f = flags[i] if flags is not None else draw_augmentation(self.flip_prob, self.rotate_prob, rng)
            src = self._stage(i, sc)
            if self.normalize_density and src.dtype == torch.float32:
                f |= ops.GRID_DENSITY
            ext.append(ops.grid_prepare(src, xb[i], R, f))
        return xb, torch.tensor(ext, dtype=torch.int32).to(self.device, non_blocking=True)
