"""Trainer parity extras: metrics, reference-format checkpoints with true resume, epoch loop."""
import os

import numpy as np
import pytest
import torch

TINY = dict(patch_size=[4] * 3, window_size=[4] * 3, embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], resolution=32,
            masking_prob=0.75, stochastic_depth_prob=0.1)


def _g16_cases():
    from oracle.gen_golden_metrics import CASES, case_inputs
    for i, (name, shape, rule) in enumerate(CASES):
        yield name, case_inputs(name, shape, rule, 1600 + i)


def _check_g16(golden, device):
    """trainer.mse / psnr / eval_metrics against golden g16 = outputs of the reference's own nerf_rpn/model/metrics.py:69-79
    (oracle/gen_golden_metrics.py), incl. the empty-mask (NaN), single-voxel and pred == target (+inf) edge cases"""
    from nerf_mae_amd.trainer import eval_metrics, mse, psnr
    g = golden("g16_metrics.npz")
    for name, (p, t, m) in _g16_cases():
        assert abs((p.double().sum() + 2 * t.double().sum()).item() - float(g[f"{name}_checksum"])) < 1e-6, "input regeneration drifted"
        assert int(m.sum()) == int(g[f"{name}_nsel"])
        p, t, m = p.to(device), t.to(device), m.to(device)
        got_m, got_p = float(mse(p, t, m)), float(psnr(p, t, m))
        np.testing.assert_allclose(got_m, float(g[f"{name}_mse"]), rtol=2e-6, equal_nan=True)
        np.testing.assert_allclose(got_p, float(g[f"{name}_psnr"]), rtol=2e-6, equal_nan=True)
        # the eval tuple path (run_swin_mae3d.py:747-760): pred / target carry 4 channels, the metric takes [..., :3]
        pad = torch.zeros(p.shape[:-1] + (1,), device=device)
        ev_p, ev_m = eval_metrics((None, None, None, torch.cat([p, pad], -1), m, torch.cat([t, pad + 0.5], -1)))
        np.testing.assert_allclose(float(ev_m), float(g[f"{name}_mse"]), rtol=2e-6, equal_nan=True)
        np.testing.assert_allclose(float(ev_p), float(g[f"{name}_psnr"]), rtol=2e-6, equal_nan=True)


def test_metrics_match_reference_golden(golden):
    _check_g16(golden, "cpu")


@pytest.mark.gpu
def test_metrics_match_reference_golden_on_device(golden):
    _check_g16(golden, "cuda")


def test_checkpoint_format_is_the_references(tmp_path):
    from nerf_mae_amd.model import SwinTransformer_MAE3D
    from nerf_mae_amd.trainer import save_checkpoint
    m = SwinTransformer_MAE3D(**TINY)
    path = os.path.join(tmp_path, "epoch_3.pt")
    save_checkpoint(path, m, 3, {"lr": 1e-4, "batch_size": 32})
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "state_dict", "train_args"} and ck["epoch"] == 3      # run_swin_mae3d.py:471-489
    m2 = SwinTransformer_MAE3D(**TINY)
    missing = m2.load_state_dict(ck["state_dict"], strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


@pytest.mark.gpu
def test_trainer_epochs_eval_best_checkpoint_and_resume(tmp_path):
    from nerf_mae_amd import data
    from nerf_mae_amd.model import SwinTransformer_MAE3D
    from nerf_mae_amd.trainer import FusedAdamW, Trainer, eval_metrics, load_checkpoint
    torch.manual_seed(0)
    model = SwinTransformer_MAE3D(compute_dtype=torch.float32, **TINY).cuda()
    train = [data.synthetic_scene((32, 32, 32), i) for i in range(4)] + [data.synthetic_scene((30, 28, 32), 9)]
    val = [data.synthetic_scene((32, 32, 32), 100 + i) for i in range(2)]
    save = os.path.join(tmp_path, "ck")
    tr = Trainer(model, train, val, batch_size=2, num_epochs=3, lr=2e-3, save_path=save, flip_prob=0.5, rotate_prob=0.5, log=lambda s: None)
    hist = tr.fit()
    assert len(hist) == 3 and all(np.isfinite(h["train_loss"]) for h in hist) and hist[-1]["train_loss"] < hist[0]["train_loss"]
    assert all("psnr" in h and np.isfinite(h["psnr"]) for h in hist)
    for f in ("model_best.pt", "epoch_1.pt", "epoch_3.pt"):
        assert os.path.exists(os.path.join(save, f))
    # metric identity: mse == loss_rgb / 3 on the same eval call
    model.eval()
    with torch.no_grad():
        xb, ext = tr.val_batcher([val[0]], flags=[0])
        out = model([xb[0]], is_eval=True)
    p, ms = eval_metrics(out)
    assert abs(float(ms) - float(out[1]) / 3.0) < 1e-5 * float(ms) and abs(float(p) + 10 * np.log10(float(ms))) < 1e-4
    # true resume: parameters AND optimizer state come back
    m2 = SwinTransformer_MAE3D(compute_dtype=torch.float32, **TINY).cuda()
    m2.flatten_parameters()
    opt2 = FusedAdamW(m2)
    ck = load_checkpoint(os.path.join(save, "epoch_3.pt"), m2, opt2)
    assert ck["epoch"] == 3 and ck["resume"]["step"] == tr.global_step and opt2.t == tr.opt.t
    assert torch.equal(m2._flat, model._flat) and torch.equal(opt2.m, tr.opt.m) and torch.equal(opt2.v, tr.opt.v)
