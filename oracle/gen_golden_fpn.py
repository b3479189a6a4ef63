"""Golden vectors for the nerf_rpn backbone drop-in, produced by the REAL reference classes in the build
container: `FPN` (nerf_rpn/model/fpn.py) and `SwinTransformer_FPN_Pretrained_Skip` (nerf_rpn/model/feature_extractor.py:1067-1187).
TEST INFRASTRUCTURE ONLY.   Run: python oracle/gen_golden_fpn.py     (needs /root/reference)

Import shims, in this process only: the torchvision 0.13.1 pieces of oracle/gen_golden.py, and an empty
`torchmetrics.JaccardIndex` name -- feature_extractor.py star-imports nerf_rpn/model/metrics.py, whose torchmetrics import is
never reached by the classes exercised here.  Weights/inputs are formula-filled, so only the reference OUTPUTS are stored."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden import _install_shims  # noqa: E402
from oracle.mae3d_oracle import formula_fill_, formula_tensor, synthetic_grid  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

FPN_CASES = [  # (tag, in_channels, out_channels, level sizes (D,H,W) fine -> coarse, batch)
    ("halves", [8, 16, 32, 64], 16, [(8, 8, 8), (4, 4, 4), (2, 2, 2), (1, 1, 1)], 2),
    ("ceil", [8, 16, 32, 64], 24, [(5, 5, 5), (3, 3, 3), (2, 2, 2), (1, 1, 1)], 1),
    ("ragged", [16, 8, 8, 24], 8, [(6, 5, 7), (3, 3, 4), (2, 2, 2), (1, 1, 1)], 1),
]


def fill_fpn_(fpn, tag):
    with torch.no_grad():
        for n, p in fpn.named_parameters():
            fan_in = p[0].numel() if p.dim() > 1 else 1
            p.copy_(formula_tensor(f"g12.{tag}.{n}", p.shape, 0.1 if p.dim() == 1 else 1.5 / np.sqrt(fan_in)))


def sample(t, n=4096):
    """checksum + strided samples of a large gradient (the full fpn_convs gradients are 4 x 1.77 M floats)"""
    f = t.detach().double().cpu().reshape(-1)
    step = max(1, f.numel() // n)
    return np.concatenate([[f.sum().item(), f.abs().sum().item()], f[::step][:n].numpy()]).astype(np.float64)


def main():
    assert os.path.isdir(REF), "reference not mounted"
    _install_shims()
    tm = types.ModuleType("torchmetrics")
    tm.JaccardIndex = object
    sys.modules["torchmetrics"] = tm
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "nerf_rpn"))
    from nerf_rpn.model import feature_extractor as FE
    from nerf_rpn.model.fpn import FPN

    torch.set_num_threads(8)
    out = {}
    # ---- the neck alone: forward + all gradients --------------------------------------------------------------------------------
    for tag, cin, cout, sizes, B in FPN_CASES:
        fpn = FPN(cin, cout, len(cin))
        fill_fpn_(fpn, tag)
        xs = [formula_tensor(f"g12.{tag}.x{i}", (B, c) + s, 1.0).requires_grad_(True) for i, (c, s) in enumerate(zip(cin, sizes))]
        ys = fpn(xs)
        loss = sum((y * formula_tensor(f"g12.{tag}.dy{i}", y.shape, 1.0)).sum() for i, y in enumerate(ys))
        loss.backward()
        for i, y in enumerate(ys):
            out[f"{tag}.y{i}"] = y.detach().numpy()
            out[f"{tag}.dx{i}"] = xs[i].grad.numpy()
        for n, p in fpn.named_parameters():
            out[f"{tag}.d_{n}"] = p.grad.numpy()
    # ---- the whole backbone: swin_s encoder at 32^3 + FPN(256), eval mode (stochastic depth off), forward + gradients ----------------
    m = FE.SwinTransformer_FPN_Pretrained_Skip(resolution=32, is_eval=True)
    formula_fill_(m)
    m.eval()
    x = torch.stack([synthetic_grid((32, 32, 32), 11), synthetic_grid((32, 32, 32), 12)])
    ys = m(x)
    loss = sum((y * formula_tensor(f"g12.skip.dy{i}", y.shape, 1.0)).sum() for i, y in enumerate(ys))
    loss.backward()
    for i, y in enumerate(ys):
        out[f"skip.y{i}"] = y.detach().numpy()
    names = [n for n, _ in m.named_parameters()]
    keep = [n for n in names if n.startswith("fpn_neck.")] + ["base.patch_partition.0.weight", "base.patch_partition.2.bias",
                                                              "base.stages.0.0.attn.qkv.weight", "base.stages.1.0.reduction.weight",
                                                              "base.stages.2.17.mlp.3.weight", "base.stages.3.2.attn.relative_position_bias_table"]
    P = dict(m.named_parameters())
    for n in keep:
        out["skip.g_" + n] = sample(P[n].grad)
    out["skip.param_names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "g12_fpn_skip.npz"), **out)
    print("wrote g12_fpn_skip.npz", os.path.getsize(os.path.join(OUT, "g12_fpn_skip.npz")), "bytes")


if __name__ == "__main__":
    main()
