"""nerf_rpn backbone drop-in: FPN neck + SwinTransformer_FPN_Pretrained_Skip.
CPU: the oracle restatement against golden vectors produced by the real reference classes (oracle/gen_golden_fpn.py);
      the product's parameter names / state_dict contract.
GPU: the HIP path (through the C ABI) against the same golden vectors and against the oracle."""
import numpy as np
import pytest
import torch

from oracle import mae3d_oracle as O
from oracle.gen_golden_fpn import FPN_CASES, fill_fpn_, sample


def relerr(a, b):
    a = torch.as_tensor(np.asarray(a)).double() if not torch.is_tensor(a) else a.detach().double().cpu()
    b = torch.as_tensor(np.asarray(b)).double() if not torch.is_tensor(b) else b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def _fpn_inputs(tag, cin, sizes, B):
    return [O.formula_tensor(f"g12.{tag}.x{i}", (B, c) + s, 1.0) for i, (c, s) in enumerate(zip(cin, sizes))]


def _skip_input():
    return torch.stack([O.synthetic_grid((32, 32, 32), 11), O.synthetic_grid((32, 32, 32), 12)])


# ---------------------------------------------------------------------------------------------------------------- CPU
@pytest.mark.parametrize("case", FPN_CASES, ids=[c[0] for c in FPN_CASES])
def test_fpn_oracle_matches_reference(golden, case):
    g = golden("g12_fpn_skip.npz")
    tag, cin, cout, sizes, B = case
    fpn = O.FPNOracle(cin, cout, len(cin))
    fill_fpn_(fpn, tag)
    xs = [x.requires_grad_(True) for x in _fpn_inputs(tag, cin, sizes, B)]
    ys = fpn(xs)
    sum((y * O.formula_tensor(f"g12.{tag}.dy{i}", y.shape, 1.0)).sum() for i, y in enumerate(ys)).backward()
    for i, y in enumerate(ys):
        assert relerr(y, g[f"{tag}.y{i}"]) < 1e-5
        assert relerr(xs[i].grad, g[f"{tag}.dx{i}"]) < 1e-5
    for n, p in fpn.named_parameters():
        assert relerr(p.grad, g[f"{tag}.d_{n}"]) < 1e-5, n


def test_skip_oracle_matches_reference(golden):
    g = golden("g12_fpn_skip.npz")
    torch.set_num_threads(8)
    m = O.FPNSkipOracle(resolution=32)
    assert [n for n, _ in m.named_parameters()] == list(g["skip.param_names"])
    O.formula_fill_(m)
    m.eval()
    ys = m(_skip_input())
    sum((y * O.formula_tensor(f"g12.skip.dy{i}", y.shape, 1.0)).sum() for i, y in enumerate(ys)).backward()
    for i, y in enumerate(ys):
        assert relerr(y, g[f"skip.y{i}"]) < 1e-4, i
    P = dict(m.named_parameters())
    for k in g.files:
        if k.startswith("skip.g_"):
            assert relerr(sample(P[k[7:]].grad), g[k]) < 1e-4, k


def test_skip_product_contract(golden):
    """parameter names equal the reference's; state_dict loads strictly both ways; deleted attributes are gone (no compute: CPU)"""
    from nerf_mae_amd.fpn import SwinTransformer_FPN_Pretrained_Skip
    g = golden("g12_fpn_skip.npz")
    m = SwinTransformer_FPN_Pretrained_Skip(resolution=32, is_eval=True)
    assert [n for n, _ in m.named_parameters()] == list(g["skip.param_names"])
    ora = O.FPNSkipOracle(resolution=32)
    r = m.load_state_dict(ora.state_dict(), strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    r = ora.load_state_dict(m.state_dict(), strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    for a in ("decoder4", "decoder3", "decoder2", "decoder1", "out", "mask_token"):
        assert not hasattr(m.base, a)
    assert m.out_channels == 256 and m.fpn_neck.in_channels == [96, 192, 384, 768]
    with pytest.raises(RuntimeError):   # no CPU fallback
        m(torch.zeros(1, 4, 32, 32, 32))


def test_skip_loads_mae_checkpoint(tmp_path):
    """the backbone is built from a checkpoint written in the MAE trainer's format (run_swin_mae3d.py:471-489)"""
    from nerf_mae_amd.fpn import SwinTransformer_FPN_Pretrained_Skip
    mae = O.build_oracle("swin_s", resolution=32)
    O.formula_fill_(mae)
    path = tmp_path / "epoch_1.pt"
    torch.save({"epoch": 1, "state_dict": mae.state_dict(), "train_args": {}}, path)
    m = SwinTransformer_FPN_Pretrained_Skip(resolution=32, checkpoint_path=str(path))
    sd = mae.state_dict()
    for k, v in m.base.state_dict().items():
        assert torch.equal(v, sd[k]), k
    with pytest.raises(AssertionError):
        SwinTransformer_FPN_Pretrained_Skip(resolution=32, checkpoint_path=str(tmp_path / "missing.pt"))


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", FPN_CASES, ids=[c[0] for c in FPN_CASES])
def test_fpn_hip_matches_reference(golden, case, dtype, tol):
    from nerf_mae_amd.fpn import FPN
    g = golden("g12_fpn_skip.npz")
    tag, cin, cout, sizes, B = case
    fpn = FPN(cin, cout, len(cin), compute_dtype=dtype)
    fill_fpn_(fpn, tag)
    fpn = fpn.cuda()
    xs = [x.cuda().requires_grad_(True) for x in _fpn_inputs(tag, cin, sizes, B)]
    ys = fpn(xs)
    assert all(y.dtype == torch.float32 and y.shape == g[f"{tag}.y{i}"].shape for i, y in enumerate(ys))
    sum((y * O.formula_tensor(f"g12.{tag}.dy{i}", y.shape, 1.0).cuda()).sum() for i, y in enumerate(ys)).backward()
    torch.cuda.synchronize()
    for i, y in enumerate(ys):
        assert relerr(y, g[f"{tag}.y{i}"]) < tol, ("y", i)
        assert relerr(xs[i].grad, g[f"{tag}.dx{i}"]) < tol, ("dx", i)
    for n, p in fpn.named_parameters():
        assert relerr(p.grad, g[f"{tag}.d_{n}"]) < tol, n


@pytest.mark.gpu
def test_skip_hip_matches_reference_fp32(golden):
    from nerf_mae_amd.fpn import SwinTransformer_FPN_Pretrained_Skip
    g = golden("g12_fpn_skip.npz")
    m = SwinTransformer_FPN_Pretrained_Skip(resolution=32, is_eval=True, compute_dtype=torch.float32)
    O.formula_fill_(m)
    m = m.cuda().eval()
    ys = m(_skip_input().cuda())
    assert [tuple(y.shape) for y in ys] == [(2, 256, 8, 8, 8), (2, 256, 4, 4, 4), (2, 256, 2, 2, 2), (2, 256, 1, 1, 1)]
    sum((y * O.formula_tensor(f"g12.skip.dy{i}", y.shape, 1.0).cuda()).sum() for i, y in enumerate(ys)).backward()
    torch.cuda.synchronize()
    for i, y in enumerate(ys):
        assert relerr(y, g[f"skip.y{i}"]) < 1e-3, i
    P = dict(m.named_parameters())
    for k in g.files:
        if k.startswith("skip.g_"):
            assert P[k[7:]].grad is not None, k
            assert relerr(sample(P[k[7:]].grad), g[k]) < 1e-3, k


@pytest.mark.gpu
def test_skip_hip_bf16_close_to_oracle():
    """bf16 compute against the fp32 oracle on the reference's own initialisation (the formula-filled weights of the golden case are
    too badly conditioned for a 24-block bf16 backward; same protocol as the MAE path in test_model_gpu.py)"""
    from nerf_mae_amd.fpn import SwinTransformer_FPN_Pretrained_Skip
    torch.manual_seed(7)
    ora = O.FPNSkipOracle(resolution=32)
    with torch.no_grad():
        for n, p in ora.named_parameters():
            if p.requires_grad and (n.endswith("bias") or "relative_position_bias_table" in n):
                p.add_(0.02 * torch.randn_like(p))
    ora.eval()
    m = SwinTransformer_FPN_Pretrained_Skip(resolution=32, is_eval=True)
    m.load_state_dict(ora.state_dict(), strict=True)
    m = m.cuda().eval()
    x = _skip_input()
    yo, yh = ora(x), m(x.cuda())
    w = [torch.randn(y.shape, generator=torch.Generator().manual_seed(i)) for i, y in enumerate(yo)]
    sum((y * wi).sum() for y, wi in zip(yo, w)).backward()
    sum((y * wi.cuda()).sum() for y, wi in zip(yh, w)).backward()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(yh, yo)):
        assert relerr(a, b) < 5e-2, i
    po = dict(ora.named_parameters())
    fa, fb, worst = [], [], (1.0, "")
    for n, p in m.named_parameters():
        if not p.requires_grad:
            continue
        a, b = p.grad.float().cpu().flatten(), po[n].grad.flatten()
        fa.append(a)
        fb.append(b)
        if b.norm() > 0:
            worst = min(worst, ((torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item(), n))
    fa, fb = torch.cat(fa), torch.cat(fb)
    assert (torch.dot(fa, fb) / (fa.norm() * fb.norm())).item() > 0.995
    assert worst[0] > 0.9, worst


@pytest.mark.gpu
def test_skip_full_size_bf16_trains():
    """BASELINE config 5 shape: swin_s encoder + FPN(256) at 160^3, bf16, one forward + backward in training mode"""
    from nerf_mae_amd.fpn import SwinTransformer_FPN_Pretrained_Skip
    torch.manual_seed(0)
    m = SwinTransformer_FPN_Pretrained_Skip(resolution=160, is_eval=True).cuda().train()
    x = torch.stack([O.synthetic_grid((160, 160, 160), 3)]).cuda()
    ys = m(x)
    assert [tuple(y.shape) for y in ys] == [(1, 256, 40, 40, 40), (1, 256, 20, 20, 20), (1, 256, 10, 10, 10), (1, 256, 5, 5, 5)]
    sum(y.square().mean() for y in ys).backward()
    torch.cuda.synchronize()
    assert all(torch.isfinite(y).all() for y in ys)
    for n, p in m.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all(), n
    # top-down pathway: level 3 feeds every finer level, so its lateral weights see gradient from all four outputs
    assert m.fpn_neck.lateral_convs[3].weight.grad.abs().sum() > 0


_FULL = {}


def _full_size_oracle():
    """BASELINE config 5 at full size on the host: FPNSkipOracle(resolution=160) forward + backward on one synthetic grid, the reference's own initialisation
    (+ small non-zero biases / bias tables, as test_skip_hip_bf16_close_to_oracle); computed once for the fp32 and the bf16 case"""
    if not _FULL:
        torch.manual_seed(11)
        torch.set_num_threads(max(1, min(64, (torch.get_num_threads() or 1) * 4)))
        ora = O.FPNSkipOracle(resolution=160)
        with torch.no_grad():
            for n, p in ora.named_parameters():
                if p.requires_grad and (n.endswith("bias") or "relative_position_bias_table" in n):
                    p.add_(0.02 * torch.randn_like(p))
        ora.eval()
        x = torch.stack([O.synthetic_grid((160, 160, 160), 5)])
        yo = ora(x)
        w = [torch.randn(y.shape, generator=torch.Generator().manual_seed(40 + i)) / y.numel() ** 0.5 for i, y in enumerate(yo)]
        sum((y * wi).sum() for y, wi in zip(yo, w)).backward()
        _FULL.update(sd={k: v.detach().clone() for k, v in ora.state_dict().items()}, x=x, w=w, y=[y.detach() for y in yo],
                     g={n: p.grad.detach().clone() for n, p in ora.named_parameters() if p.grad is not None})
    return _FULL


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol,cos_all,cos_each", [(torch.float32, 1e-3, 0.9999, 0.999), (torch.bfloat16, 5e-2, 0.995, 0.9)], ids=["fp32", "bf16"])
def test_skip_full_size_matches_oracle(dtype, tol, cos_all, cos_each):
    """BASELINE config 5 at FULL size (feature_extractor.py:1176-1187, fpn.py:135-185): swin_s encoder + FPN(256) on one 160^3 grid against the oracle --
    the four NCDHW maps (40^3 ... 5^3 x 256) to `tol` of their max-norm, every parameter gradient by cosine"""
    from nerf_mae_amd.fpn import SwinTransformer_FPN_Pretrained_Skip
    ref = _full_size_oracle()
    m = SwinTransformer_FPN_Pretrained_Skip(resolution=160, is_eval=True, compute_dtype=dtype)
    m.load_state_dict(ref["sd"], strict=True)
    m = m.cuda().eval()
    yh = m(ref["x"].cuda())
    assert [tuple(y.shape) for y in yh] == [(1, 256, 40, 40, 40), (1, 256, 20, 20, 20), (1, 256, 10, 10, 10), (1, 256, 5, 5, 5)]
    sum((y * wi.cuda()).sum() for y, wi in zip(yh, ref["w"])).backward()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(yh, ref["y"])):
        assert relerr(a, b) < tol, ("map", i, relerr(a, b))
    fa, fb, rows = [], [], []
    for n, p in m.named_parameters():
        if not p.requires_grad or n not in ref["g"]:
            continue
        a, b = p.grad.float().cpu().flatten(), ref["g"][n].flatten()
        fa.append(a)
        fb.append(b)
        if b.norm() > 0:
            rows.append(((torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item(), n, a.norm().item(), b.norm().item()))
    rows.sort()
    fa, fb = torch.cat(fa), torch.cat(fb)
    assert rows[0][0] > cos_each, rows[:8]
    assert (torch.dot(fa, fb) / (fa.norm() * fb.norm())).item() > cos_all, rows[:8]
