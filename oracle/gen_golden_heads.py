"""Golden vectors for the voxel super-resolution / voxel semantics heads, produced by the REAL reference classes
`SwinTransformer_VoxelSR_Pretrained_Skip` and `SwinTransformer_VoxelSemantics_Pretrained_Skip` (nerf_rpn/model/feature_extractor.py:
1898-2244, 2521-2848) in the build container.  TEST INFRASTRUCTURE ONLY.   Run: python oracle/gen_golden_heads.py  (needs /root/reference)

The classes hard-code the swin_s backbone; the fixture runs them at resolution 32 (token grid 8^3) with the reference's initialisation distributions seeded per
parameter (oracle.seeded_reference_init_: formula-filled weights put the 2^3-voxel InstanceNorms of the coarsest decoder level in an
ill-conditioned regime where two correct fp32 implementations differ by percents), so only inputs' seeds and the reference OUTPUTS are stored: prediction checksums + strided samples, the loss terms, and checksums + samples of
the gradients of every parameter group.  VoxelSR: `nn.Upsample(scale_factor=1.6)` turns 32^3 into 51^3; the target is padded to 51^3
(the class's `output_resolution`, 256 for a 160^3 input, is set to 51 on the instance -- the same arithmetic at a size that fits a test)."""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden import _install_shims  # noqa: E402
from oracle.mae3d_oracle import formula_tensor, seeded_reference_init_, synthetic_grid  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
R, RO, K, SEED = 32, 51, 19, 5


def inputs():
    return [synthetic_grid((32, 32, 32), 601), synthetic_grid((30, 32, 27), 602)]


def sr_targets():
    """the same scenes 'rendered' at the output resolution: synthetic grids of 51^3 and 48x51x43"""
    return [synthetic_grid((51, 51, 51), 611), synthetic_grid((48, 51, 43), 612)]


def sem_labels():
    """semantic label grids (1, W, L, H): class ids 0..18, 0 = unlabelled (about 40 %)"""
    out = []
    for shape, seed in (((32, 32, 32), 621), ((30, 32, 27), 622)):
        g = torch.Generator().manual_seed(seed)
        lab = torch.randint(1, K, shape, generator=g).float()
        lab[torch.rand(shape, generator=g) < 0.4] = 0.0
        out.append(lab[None])
    return out


def class_weights():
    return formula_tensor("g15.class_weights", (K,), 0.4, 1.0)


def sample(t, n=2048):
    f = t.detach().double().cpu().reshape(-1)
    step = max(1, f.numel() // n)
    return np.concatenate([[f.sum().item(), f.abs().sum().item()], f[::step][:n].numpy()]).astype(np.float64)


def grad_summary(m):
    """per top-level group: [sum, abs-sum, L2] of all gradients + samples of a few new-head tensors"""
    groups = {}
    for n, p in m.named_parameters():
        if p.grad is None:
            continue
        top = n.split(".")[0] if not n.startswith("base.") else ".".join(n.split(".")[:2])
        a = groups.setdefault(top, [0.0, 0.0, 0.0])
        g = p.grad.double()
        a[0] += g.sum().item(); a[1] += g.abs().sum().item(); a[2] += (g * g).sum().item()
    names = sorted(groups)
    return np.array(names), np.array([groups[k] for k in names], np.float64)


def main():
    assert os.path.isdir(REF), "reference not mounted"
    _install_shims()
    tm = types.ModuleType("torchmetrics")
    tm.JaccardIndex = object
    sys.modules["torchmetrics"] = tm
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "nerf_rpn"))
    from nerf_rpn.model import feature_extractor as FE
    torch.set_num_threads(8)
    out = {}
    with contextlib.redirect_stdout(io.StringIO()):   # the reference prints banners and tensor shapes
        sr = FE.SwinTransformer_VoxelSR_Pretrained_Skip(resolution=R, out_resolution=256, is_eval=True)
    sr.output_resolution = RO
    seeded_reference_init_(sr, SEED)
    sr.train()
    for mod in sr.modules():      # stochastic depth off: the fixture is deterministic
        if mod.__class__.__name__ == "StochasticDepth":
            mod.p = 0.0
    with contextlib.redirect_stdout(io.StringIO()):
        pred = sr(inputs())
    assert tuple(pred.shape) == (2, 4, RO, RO, RO), pred.shape
    loss = sr.loss_fn(sr_targets(), pred)
    loss.backward()
    out["sr_pred"] = sample(pred)
    out["sr_loss"] = np.array([loss.item()])
    out["sr_grad_names"], out["sr_grad_sums"] = grad_summary(sr)
    for n in ("encoder1.layer.conv1.weight", "encoder1.layer.conv3.weight", "decoder1.conv_block.conv1.weight", "decoder1.transp_conv.weight", "voxel_out.conv.weight",
              "voxel_out.conv.bias", "base.decoder2.conv_block.conv2.weight", "base.stages.0.0.attn.qkv.weight", "base.patch_partition.0.weight"):
        out["sr_g." + n] = sample(dict(sr.named_parameters())[n].grad, 512)
    print("VoxelSR: loss %.6f, pred |sum| %.4f" % (loss.item(), out["sr_pred"][1]))

    with contextlib.redirect_stdout(io.StringIO()):
        se = FE.SwinTransformer_VoxelSemantics_Pretrained_Skip(resolution=R, out_channels=K, is_eval=True, class_weights=class_weights())
    seeded_reference_init_(se, SEED)
    se.train()
    for mod in se.modules():
        if mod.__class__.__name__ == "StochasticDepth":
            mod.p = 0.0
    pred = se(inputs())
    assert tuple(pred.shape) == (2, K, R, R, R)
    loss, ce, iou = se.loss_fn(sem_labels(), pred)
    loss.backward()
    out["sem_pred"] = sample(pred)
    out["sem_loss"] = np.array([loss.item(), ce.item(), iou.item()])
    out["sem_grad_names"], out["sem_grad_sums"] = grad_summary(se)
    for n in ("encoder1.layer.conv2.weight", "decoder1.conv_block.conv3.weight", "sem_out.conv.weight", "sem_out.conv.bias", "base.decoder3.transp_conv.weight"):
        out["sem_g." + n] = sample(dict(se.named_parameters())[n].grad, 512)
    print("VoxelSemantics: loss %.6f iou %.6f" % (loss.item(), iou.item()))
    np.savez_compressed(os.path.join(OUT, "g15_voxel_heads.npz"), **out)
    print("wrote g15_voxel_heads.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
