"""Voxel super-resolution / voxel semantics heads (nerf_rpn/model/feature_extractor.py:1898-2244, 2521-2848).
CPU: the oracle restatement (oracle/heads_oracle.py) against golden g15 produced by the REAL reference classes.  GPU: the HIP heads
(nerf_mae_amd.heads) against the golden (fp32) and the oracle (bf16), state_dict exchange, a full-size training step."""
import os

import numpy as np
import pytest
import torch

from oracle import heads_oracle as HO
from oracle import mae3d_oracle as O
from oracle.gen_golden_heads import K, R, RO, SEED, class_weights, grad_summary, inputs, sample, sem_labels, sr_targets


def _chk(got, want, rtol, name):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    scale = np.abs(want).max() + 1e-30
    err = np.abs(got - want).max() / scale
    assert err < rtol, f"{name}: max rel err {err:.3e} (tol {rtol})"


def _build_oracles():
    torch.manual_seed(0)
    sr = HO.VoxelSROracle(resolution=R, out_resolution=RO, stochastic_depth_prob=0.0)
    sr.scale = 1.6
    O.seeded_reference_init_(sr, SEED)
    se = HO.VoxelSemanticsOracle(resolution=R, out_channels=K, class_weights=class_weights(), stochastic_depth_prob=0.0)
    O.seeded_reference_init_(se, SEED)
    return sr.train(), se.train()


def test_heads_oracle_matches_reference_golden(golden):
    g = golden("g15_voxel_heads.npz")
    torch.set_num_threads(8)
    sr, se = _build_oracles()
    pred = sr(inputs())
    assert tuple(pred.shape) == (2, 4, RO, RO, RO)
    loss = sr.forward_loss(sr_targets(), pred)
    loss.backward()
    _chk(sample(pred), g["sr_pred"], 2e-5, "sr pred")
    _chk([loss.item()], g["sr_loss"], 2e-5, "sr loss")
    names, sums = grad_summary(sr)
    assert list(names) == list(g["sr_grad_names"])
    _chk(sums, g["sr_grad_sums"], 2e-3, "sr grad group sums")
    for k in g.files:
        if k.startswith("sr_g."):
            _chk(sample(dict(sr.named_parameters())[k[5:]].grad, 512), g[k], 2e-3, k)
    pred = se(inputs())
    loss, ce, iou = se.forward_loss(sem_labels(), pred)
    loss.backward()
    _chk(sample(pred), g["sem_pred"], 2e-5, "sem pred")
    _chk([loss.item(), ce.item(), iou.item()], g["sem_loss"], 2e-5, "sem loss")
    names, sums = grad_summary(se)
    assert list(names) == list(g["sem_grad_names"])
    _chk(sums, g["sem_grad_sums"], 2e-3, "sem grad group sums")
    for k in g.files:
        if k.startswith("sem_g."):
            _chk(sample(dict(se.named_parameters())[k[6:]].grad, 512), g[k], 2e-3, k)


def _hip_heads(dtype):
    from nerf_mae_amd.heads import SwinTransformer_VoxelSemantics_Pretrained_Skip, SwinTransformer_VoxelSR_Pretrained_Skip
    sr_o, se_o = _build_oracles()
    sr = SwinTransformer_VoxelSR_Pretrained_Skip(resolution=R, out_resolution=256, is_eval=True, compute_dtype=dtype)
    sr.load_state_dict(sr_o.state_dict(), strict=True)       # the oracle's keys are the reference's (pinned by the golden generator)
    se = SwinTransformer_VoxelSemantics_Pretrained_Skip(resolution=R, out_channels=K, is_eval=True, class_weights=class_weights(), compute_dtype=dtype)
    se.load_state_dict(se_o.state_dict(), strict=True)
    for m in (sr, se):
        m.cuda().train()
        for blk in m.base.modules():
            if hasattr(blk, "sd_prob"):
                blk.sd_prob = 0.0
    return sr, se, sr_o, se_o


@pytest.mark.gpu
def test_hip_heads_match_reference_golden_fp32(golden):
    """fp32 HIP heads against golden g15 from the reference classes: predictions, losses, soft IoU, gradients of every parameter group"""
    g = golden("g15_voxel_heads.npz")
    sr, se, _, _ = _hip_heads(torch.float32)
    xs = [t.cuda() for t in inputs()]
    pred = sr(xs)
    assert tuple(pred.shape) == (2, 4, RO, RO, RO) and pred.dtype == torch.float32
    loss = sr.loss_fn([t.cuda() for t in sr_targets()], pred)
    loss.backward()
    torch.cuda.synchronize()
    _chk(sample(pred), g["sr_pred"], 1e-3, "sr pred")
    _chk([loss.item()], g["sr_loss"], 1e-4, "sr loss")
    names, sums = grad_summary(sr)
    assert list(names) == list(g["sr_grad_names"])
    _chk(sums[:, 2], g["sr_grad_sums"][:, 2], 5e-3, "sr grad group norms")
    for k in g.files:
        if k.startswith("sr_g."):
            _chk(sample(dict(sr.named_parameters())[k[5:]].grad, 512)[2:], g[k][2:], 5e-3, k)
    pred = se(xs)
    assert tuple(pred.shape) == (2, K, R, R, R)
    loss, ce, iou = se.loss_fn([t.cuda() for t in sem_labels()], pred)
    loss.backward()
    torch.cuda.synchronize()
    _chk(sample(pred), g["sem_pred"], 1e-3, "sem pred")
    _chk([loss.item(), ce.item(), iou.item()], g["sem_loss"], 2e-4, "sem loss / iou")
    names, sums = grad_summary(se)
    _chk(sums[:, 2], g["sem_grad_sums"][:, 2], 5e-3, "sem grad group norms")
    for k in g.files:
        if k.startswith("sem_g."):
            _chk(sample(dict(se.named_parameters())[k[6:]].grad, 512)[2:], g[k][2:], 5e-3, k)


@pytest.mark.gpu
def test_hip_heads_bf16_track_the_oracle():
    sr, se, sr_o, se_o = _hip_heads(torch.bfloat16)
    xs = inputs()
    po = sr_o(xs)
    lo = sr_o.forward_loss(sr_targets(), po)
    ph = sr([t.cuda() for t in xs])
    lh = sr.loss_fn([t.cuda() for t in sr_targets()], ph)
    lh.backward()
    assert abs(lh.item() - lo.item()) / abs(lo.item()) < 3e-2
    _chk(ph.detach().float().cpu().numpy(), po.detach().numpy(), 8e-2, "bf16 sr pred")
    po = se_o(xs)
    lo = se_o.forward_loss(sem_labels(), po)
    ph = se([t.cuda() for t in xs])
    lh = se.loss_fn([t.cuda() for t in sem_labels()], ph)
    lh[0].backward()
    assert abs(lh[0].item() - lo[0].item()) / abs(lo[0].item()) < 3e-2
    assert abs(lh[2].item() - lo[2].item()) < 5e-3
    assert all(torch.isfinite(p.grad).all() for p in se.parameters() if p.grad is not None)


@pytest.mark.gpu
def test_voxel_sr_full_size_training_step_160_to_256():
    """the reference's configuration: 160^3 input, 256^3 output (nn.Upsample(scale_factor=1.6)), bf16, one forward + backward"""
    from nerf_mae_amd.heads import SwinTransformer_VoxelSR_Pretrained_Skip
    torch.manual_seed(0)
    m = SwinTransformer_VoxelSR_Pretrained_Skip(resolution=160, out_resolution=256, is_eval=True).cuda().train()
    x = [O.synthetic_grid((160, 132, 96), 5).cuda()]
    tgt = [O.synthetic_grid((256, 211, 154), 6).cuda()]
    pred = m(x)
    assert tuple(pred.shape) == (1, 4, 256, 256, 256)
    loss = m.loss_fn(tgt, pred)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and loss.item() > 0
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None))
    assert torch.isfinite(gn) and gn.item() > 0
    # nearest upsampling: every output voxel equals the head output at its source voxel (index rule of nn.Upsample(scale_factor=1.6))
    src = (torch.arange(256, dtype=torch.float32) * torch.tensor(1.0 / 1.6, dtype=torch.float32)).floor().long().clamp(max=159).cuda()
    small = pred[0, :, ::8, ::8, ::8]                     # output voxels 0, 8, 16, ... read source voxels 0, 5, 10, ...: compare two routes
    again = pred[0][:, src.new_tensor([0, 8, 16])][:, :, src.new_tensor([0, 8, 16])][:, :, :, src.new_tensor([0, 8, 16])]
    assert torch.equal(small[:, :3, :3, :3], again)
