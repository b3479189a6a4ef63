"""TEST INFRASTRUCTURE -- generates tests/golden/g16_metrics.npz from the REAL reference in the build container.

Imports /root/reference/nerf_rpn/model/metrics.py (mse / psnr, :69-79 -- what run_swin_mae3d.py:758-760 calls on the eval tuple)
with stub modules for the two imports the container lacks (`torchmetrics.JaccardIndex`, `matplotlib.pyplot`: neither is touched by
mse / psnr), runs it on seeded inputs shaped like the eval tuple `(pred[..., :3], target[..., :3], mask)` of
`SwinTransformer_MAE3D.forward(is_eval=True)` and stores inputs' seeds + outputs.  Cases: dense mask, sparse mask, one selected voxel,
EMPTY mask (the reference returns NaN: mean of an empty selection), a bool mask that needs broadcasting over the channel axis, and a
pred == target case (mse 0 -> psnr +inf).  The inputs themselves are regenerated from the seeds by the test (torch.manual_seed on the
CPU generator is stable across the container and the GPU box: same torch build).

    python oracle/gen_golden_metrics.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden", "g16_metrics.npz")

CASES = [  # name, shape of the patchified RGB tensors (B,g,g,g,64,3), mask rule
    ("dense", (2, 3, 3, 3, 64, 3), "gt0.4"),
    ("sparse", (1, 2, 2, 2, 64, 3), "gt0.97"),
    ("single", (1, 2, 2, 2, 64, 3), "single"),
    ("empty", (1, 2, 2, 2, 64, 3), "empty"),
    ("equal", (1, 2, 2, 2, 64, 3), "equal"),
    ("big", (2, 8, 8, 8, 64, 3), "gt0.7"),
]


def case_inputs(name, shape, rule, seed):
    """seeded inputs of one case (shared by the generator and the test)"""
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(shape, generator=g)
    t = torch.rand(shape, generator=g)
    u = torch.rand(shape[:-1] + (1,), generator=g)
    if rule.startswith("gt"):
        m = u > float(rule[2:])
    elif rule == "single":
        m = torch.zeros_like(u, dtype=torch.bool)
        m.view(-1)[17] = True
    elif rule == "empty":
        m = torch.zeros_like(u, dtype=torch.bool)
    else:  # equal
        m = u > 0.5
        p = t.clone()
    return p, t, m


def _load_reference_metrics():
    tm = types.ModuleType("torchmetrics")
    tm.JaccardIndex = object
    mpl = types.ModuleType("matplotlib")
    plt = types.ModuleType("matplotlib.pyplot")
    mpl.pyplot = plt
    saved = {k: sys.modules.get(k) for k in ("torchmetrics", "matplotlib", "matplotlib.pyplot")}
    sys.modules.update({"torchmetrics": tm, "matplotlib": mpl, "matplotlib.pyplot": plt})
    try:
        spec = importlib.util.spec_from_file_location("_ref_metrics", os.path.join(REF, "nerf_rpn", "model", "metrics.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def main():
    ref = _load_reference_metrics()
    out = {}
    for i, (name, shape, rule) in enumerate(CASES):
        p, t, m = case_inputs(name, shape, rule, 1600 + i)
        with np.errstate(all="ignore"):
            out[f"{name}_mse"] = np.float64(ref.mse(p, t, m).item())
            out[f"{name}_psnr"] = np.float64(ref.psnr(p, t, valid_mask=m).item())
        out[f"{name}_nsel"] = np.int64(int(m.sum()))
        out[f"{name}_checksum"] = np.float64((p.double().sum() + 2 * t.double().sum()).item())   # guards the input regeneration
        print(f"g16 {name:7s} nsel {int(m.sum()):6d} mse {out[f'{name}_mse']:.9g} psnr {out[f'{name}_psnr']:.9g}")
    np.savez(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
