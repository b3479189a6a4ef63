"""GPU: whole hot path (SwinTransformer_MAE3D forward + backward through the C ABI) vs the CPU oracle on identical
weights (strict state_dict transplant), inputs and masks.  fp32 mode must meet the north-star 1e-3 relative bar."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


from tests._metrics import assert_close, relerr  # noqa: E402

# bf16 whole-model outputs: max-norm and relative L2 within 3e-2 (measured 1.0-1.6e-2); the elementwise bound |a-b| <= t (|b| + rms(b)) holds
# at t = 0.2 (measured worst 0.11-0.12): the absolute error of a bf16 network output is set by the magnitude of the 48-term dot products
# and 160^3-voxel InstanceNorms that feed it -- uniform over the tensor -- not by the element's own magnitude
BF16_ELEM_MULT = 0.2 / 3e-2


def _pair(cfg, dtype, res=32, sd=0.0, mask_p=0.75, init="formula"):
    from nerf_mae_amd.model import SwinTransformer_MAE3D
    from oracle import mae3d_oracle as O
    torch.manual_seed(1234)
    ora = O.MAE3DOracle(resolution=res, masking_prob=mask_p, stochastic_depth_prob=sd, pad_pos_embed=cfg['embed_dim'] % 6 != 0, **cfg)
    if init == "formula":
        O.formula_fill_(ora)
    else:  # the reference's own initialisation (trunc_normal .02 linears, kaiming convs) + non-trivial biases/bias tables
        with torch.no_grad():
            for n, p in ora.named_parameters():
                if p.requires_grad and (n.endswith("bias") or "relative_position_bias_table" in n):
                    p.add_(0.02 * torch.randn_like(p))
    hip = SwinTransformer_MAE3D(patch_size=[4] * 3, window_size=[4] * 3, resolution=res, masking_prob=mask_p, stochastic_depth_prob=sd,
                                compute_dtype=dtype, **cfg)
    missing = hip.load_state_dict(ora.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return ora, hip.cuda()


TINY = dict(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8])
SWIN_T = dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24])
SWIN_B_WIDTH = dict(embed_dim=128, depths=[2, 2, 2, 2], num_heads=[4, 8, 16, 32])   # BASELINE config 4 width (defined deviation)


def _run_both(ora, hip, xs, seed):
    from oracle import mae3d_oracle as O
    g = ora.resolution // 4
    bm = O.draw_block_mask((g, g, g), ora.masking_prob, rng=random.Random(seed))
    lo = ora(xs, block_mask=bm, return_pred=True)
    lo[0].backward()
    hip.zero_grad()
    lh = hip([t.cuda() for t in xs], block_mask=bm, return_pred=True)
    lh[0].backward()
    torch.cuda.synchronize()
    return lo, lh


def _grads_vs_fp64(ora, hip, xs, seed):
    """returns loss/pred tuples and per-parameter (hip error, reference-fp32 error) measured against an fp64 oracle"""
    import copy
    from oracle import mae3d_oracle as O
    lo, lh = _run_both(ora, hip, xs, seed)
    o64 = copy.deepcopy(ora).double()
    o64.zero_grad()
    g = ora.resolution // 4
    bm = O.draw_block_mask((g, g, g), ora.masking_prob, rng=random.Random(seed))
    l64 = o64([t.double() for t in xs], block_mask=bm.double(), return_pred=True)
    l64[0].backward()
    p64, po, ph = dict(o64.named_parameters()), dict(ora.named_parameters()), dict(hip.named_parameters())
    rows = {}
    for n, p in p64.items():
        if p.grad is None:
            continue
        if n.endswith(("conv1.bias", "conv2.bias", "conv3.bias")):
            # bias feeding an InstanceNorm: gradient identically zero in exact arithmetic (rounding noise in the reference,
            # exactly 0 here because the bias add is skipped) -- DESIGN.md "conv biases"
            assert ph[n].grad.abs().max().item() == 0.0 and p.grad.abs().max().item() < 1e-9  # fp64 truth is ~0
            continue
        a, b = ph[n].grad.float().cpu().flatten(), p.grad.float().flatten()
        rows[n] = (relerr(a, b), relerr(po[n].grad, p.grad), (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item())
    return lo, lh, l64, rows


def test_fp32_formula_weights_forward_matches_oracle():
    """north-star bar: reconstructed grids within 1e-3 relative of the reference fp32 path on identical inputs
    (formula-filled weights = the same tensors the golden fixtures pin the oracle with)."""
    from oracle import mae3d_oracle as O
    for cfg in (TINY, SWIN_T, SWIN_B_WIDTH):
        ora, hip = _pair(cfg, torch.float32)
        xs = [O.synthetic_grid((32, 32, 32), 11), O.synthetic_grid((30, 28, 32), 12)]   # second sample exercises pad_tensor
        lo, lh = _run_both(ora, hip, xs, 42)
        for a, b, n in zip(lh[:3], lo[:3], ("loss", "loss_rgb", "loss_alpha")):
            assert abs(a.item() - b.item()) / abs(b.item()) < 1e-4, (n, a.item(), b.item())
        # formula-filled weights (the golden fixtures' tensors) put the 2^3-voxel InstanceNorms of the coarsest decoder level into a regime where
        # two correct fp32 implementations differ far more than with the reference's initialisation (the next test: 2e-6 there): max-norm and
        # relative L2 hold the north-star 1e-3; the elementwise bound (atol = rtol * rms) is met at 2e-3 (measured 1.12e-3)
        assert_close(lh[3], lo[3], 1e-3, "reconstructed grid", elem_mult=2.0)


@pytest.mark.parametrize("cfg,name,res", [(SWIN_T, "swin_t", 32), (TINY, "tiny", 96)])
def test_fp32_forward_backward_matches_oracle(cfg, name, res):
    """reference-style initialisation; gradients are judged against an fp64 oracle: the HIP fp32 path must be as close to
    it as the reference's own fp32 arithmetic is (the gradient map amplifies 1e-7 noise ~1e3-1e4x)."""
    from oracle import mae3d_oracle as O
    ora, hip = _pair(cfg, torch.float32, res=res, init="default")
    xs = [O.synthetic_grid((res, res, res), 11), O.synthetic_grid((res - 2, res - 4, res), 12)]
    lo, lh, l64, rows = _grads_vs_fp64(ora, hip, xs, 42)
    for a, b in zip(lh[:3], lo[:3]):
        assert abs(a.item() - b.item()) / abs(b.item()) < 1e-5
    assert_close(lh[3], lo[3], 1e-4, "reconstructed grid")
    worst = max(rows.items(), key=lambda kv: kv[1][0])
    print(f"[{name}] worst grad err vs fp64: hip {worst[1][0]:.2e} (reference fp32 {worst[1][1]:.2e}) at {worst[0]}")
    ref_noise = max(r[1] for r in rows.values())
    for n, (eh, er, cos) in rows.items():
        # fp32 atomics make the summation order (hence the amplified rounding noise) vary run to run: observed 3e-4 .. 2.5e-3
        assert eh < max(5e-3, 10 * ref_noise), f"{name} grad {n}: {eh:.3e} vs reference-fp32 noise {ref_noise:.3e}"
        assert cos > 0.9999, (n, cos)


def test_swin_b_width_trains_in_bf16():
    """embed_dim 128 / heads [4,8,16,32] (64-channel last decoder level: the generic conv kernels instead of the 48-channel ones)"""
    from oracle import mae3d_oracle as O
    ora, hip = _pair(SWIN_B_WIDTH, torch.bfloat16, res=32, init="default")
    xs = [O.synthetic_grid((32, 32, 32), 5), O.synthetic_grid((32, 28, 30), 6)]
    lo, lh = _run_both(ora, hip, xs, 11)
    assert abs(lh[0].item() - lo[0].item()) / abs(lo[0].item()) < 2e-2
    assert_close(lh[3], lo[3], 3e-2, "reconstructed grid", elem_mult=BF16_ELEM_MULT)
    fa = torch.cat([p.grad.float().cpu().flatten() for n, p in hip.named_parameters() if p.grad is not None and p.requires_grad])
    fb = torch.cat([dict(ora.named_parameters())[n].grad.flatten() for n, p in hip.named_parameters() if p.grad is not None and p.requires_grad])
    assert (torch.dot(fa, fb) / (fa.norm() * fb.norm())).item() > 0.995


def test_fused_and_unfused_decoder_tail_agree():
    from oracle import mae3d_oracle as O
    ora, hip = _pair(TINY, torch.float32, res=32, init="default")
    xs = [O.synthetic_grid((32, 32, 32), 11).cuda(), O.synthetic_grid((30, 28, 32), 12).cuda()]
    bm = O.draw_block_mask((8, 8, 8), ora.masking_prob, rng=random.Random(3))
    grads = []
    for fuse in (True, False):
        hip.fuse_tail = fuse
        hip.zero_grad()
        out = hip(xs, block_mask=bm)
        out[0].backward()
        grads.append((out[0].item(), hip._flat_grad.clone()))
    assert abs(grads[0][0] - grads[1][0]) < 1e-6 * abs(grads[1][0])   # (fp64 atomics: not bit-reproducible run to run)
    assert relerr(grads[0][1], grads[1][1]) < 5e-3   # same conditioning caveat as above: 1e-7 differences are amplified
    a, b = grads[0][1].double(), grads[1][1].double()
    assert (torch.dot(a, b) / (a.norm() * b.norm())).item() > 0.99999


def test_stage_flush_behind_the_first_chain_kernel_changes_nothing(monkeypatch):
    """ops.WQ_FLUSH_AFTER_FIRST only moves the POINT at which a stage's queued weight gradients fork off (behind the next stage's first input-gradient
    kernel instead of in front of it): same launches, same operands -- every gradient but the few that end in order-dependent atomics is bit-identical."""
    from nerf_mae_amd import ops
    from oracle import mae3d_oracle as O
    ora, hip = _pair(SWIN_T, torch.bfloat16, res=32, init="default")
    xs = [O.synthetic_grid((32, 32, 32), 21).cuda(), O.synthetic_grid((32, 30, 27), 22).cuda()]
    bm = O.draw_block_mask((8, 8, 8), ora.masking_prob, rng=random.Random(5))
    grads = []
    for flag in (False, True, False):
        monkeypatch.setattr(ops, "WQ_FLUSH_AFTER_FIRST", flag)
        hip.zero_grad()
        out = hip(xs, block_mask=bm)
        out[0].backward()
        torch.cuda.synchronize()
        assert not hip._wq._due and not hip._wq.pending and not hip._wq.deferred
        grads.append((out[0].item(), hip._flat_grad.clone()))
    noise = relerr(grads[0][1], grads[2][1])            # run-to-run difference of the same setting (atomics)
    assert abs(grads[0][0] - grads[1][0]) <= 1e-6 * abs(grads[0][0])
    assert relerr(grads[1][1], grads[0][1]) <= max(4 * noise, 1e-6), (relerr(grads[1][1], grads[0][1]), noise)
    assert torch.isfinite(grads[1][1]).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_patch_embed_of_the_kept_tokens_only_changes_nothing(monkeypatch, dtype):
    """ops.EMBED_KEPT: im2row, GEMM, LayerNorm and weight gradient of the patch embed on the compact rows of the kept tokens (swin_mae3d.py:1375-1380 replaces
    the others by mask_token).  Against the full pass: the same loss, and gradients that differ by the order of fp32 sums only -- with a ragged second sample,
    with a capacity larger than the kept count (zeroed tail rows), with one block removed and with every block removed"""
    from nerf_mae_amd import ops
    from nerf_mae_amd.model import embed_capacity_rows
    from oracle import mae3d_oracle as O
    ora, hip = _pair(SWIN_T, dtype, res=32, init="default")
    xs = [O.synthetic_grid((32, 32, 32), 41).cuda(), O.synthetic_grid((32, 30, 27), 42).cuda()]
    one = torch.zeros(8, 8, 8, dtype=torch.uint8)
    one[4:, :4, 4:] = 1   # one block removed (with none removed the reference's loss_alpha is 0 / 0)
    masks = [O.draw_block_mask((8, 8, 8), 0.75, rng=random.Random(8)), one, torch.ones(8, 8, 8, dtype=torch.uint8)]
    assert embed_capacity_rows(8, kept=130) == 192 and embed_capacity_rows(8, kept=0) == 64 and embed_capacity_rows(40, p_remove=0.75) < 0.4 * 64000
    for bm in masks:
        res = []
        for flag in (False, True, False):
            monkeypatch.setattr(ops, "EMBED_KEPT", flag)
            hip.zero_grad()
            out = hip(xs, block_mask=bm)
            out[0].backward()
            torch.cuda.synchronize()
            res.append((out[0].item(), hip._flat_grad.clone()))
        noise = relerr(res[0][1], res[2][1])
        assert abs(res[0][0] - res[1][0]) <= 5e-6 * abs(res[0][0]), (res[0][0], res[1][0])   # (run-to-run: the InstanceNorm statistics are sums of fp64 atomics of fp32 partials)
        assert relerr(res[1][1], res[0][1]) <= max(4 * noise, 2e-6 if dtype == torch.float32 else 2e-3), (relerr(res[1][1], res[0][1]), noise)
        assert torch.isfinite(res[1][1]).all()
    # a capacity above the kept count (what a captured step runs with): forward_static with model._embed_cap set
    monkeypatch.setattr(ops, "EMBED_KEPT", True)
    bm = masks[0]
    xb, ext = hip.transform(xs, torch.device("cuda"))
    md = bm.to(torch.uint8).contiguous().view(-1).cuda()
    outs = []
    for cap in (None, 512, embed_capacity_rows(8, kept=int(512 - int(bm.sum())))):
        hip._embed_cap = cap
        try:
            hip.zero_grad()
            torch.manual_seed(3)
            l = hip.forward_static(xb, ext, md)
            l[0].backward()
            torch.cuda.synchronize()
            outs.append((l[0].item(), hip._flat_grad.clone()))
        finally:
            hip._embed_cap = None
    for o in outs[1:]:
        assert abs(o[0] - outs[0][0]) <= 5e-6 * abs(outs[0][0])
        assert relerr(o[1], outs[0][1]) <= (2e-5 if dtype == torch.float32 else 5e-3)


def test_centered_decoder1_agrees_with_the_classic_form(monkeypatch):
    """ops.CCONV_CENTERED (csrc/cconv.hip): conv1 -> InstanceNorm -> LeakyReLU -> conv2 of decoder1 with the mean taken from the coarse tensor, z = lrelu(y1 - mean)
    stored, 1 / std in conv2's weights.  The same function in real arithmetic, different rounding points (rstd * bf16(z) vs bf16(rstd * t)): loss and gradients agree
    to bf16 accuracy with the classic form, at two samples per step (8 % B == 0: the scaled weight-gradient reduce) with one ragged"""
    from nerf_mae_amd import ops
    from oracle import mae3d_oracle as O
    ora, hip = _pair(SWIN_T, torch.bfloat16, res=64, init="default")
    xs = [O.synthetic_grid((64, 64, 64), 51).cuda(), O.synthetic_grid((64, 60, 51), 52).cuda()]
    bm = O.draw_block_mask((16, 16, 16), ora.masking_prob, rng=random.Random(9))
    calls = []
    real = ops.cconv_fwd_centered
    monkeypatch.setattr(ops, "cconv_fwd_centered", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    res = {}
    for flag in (True, False):   # (the packer allocates the mean table at the first forward: the flag is on there)
        monkeypatch.setattr(ops, "CCONV_CENTERED", flag)
        n0 = len(calls)
        hip.zero_grad()
        out = hip(xs, block_mask=bm, return_pred=True)
        out[0].backward()
        torch.cuda.synchronize()
        assert len(calls) - n0 == (1 if flag else 0)   # the form under test is the one that ran
        res[int(flag)] = (out[0].item(), hip._flat_grad.clone(), out[3].clone())
    assert abs(res[0][0] - res[1][0]) <= 2e-3 * abs(res[0][0]), (res[0][0], res[1][0])
    assert_close(res[1][2], res[0][2].cpu(), 3e-2, "reconstructed grid, centered vs classic", elem_mult=BF16_ELEM_MULT)
    a, b = res[1][1].double(), res[0][1].double()
    assert (torch.dot(a, b) / (a.norm() * b.norm())).item() > 0.999
    # per parameter: the decoder1 / head gradients are the ones the change touches
    rows = []
    for n, p in hip.named_parameters():
        if n.startswith(("decoder1.", "out.")) and p.requires_grad:
            off = hip._offsets[id(p)]
            ga, gb = res[1][1][off:off + p.numel()].double(), res[0][1][off:off + p.numel()].double()
            if gb.norm() > 0:   # (conv biases in front of the affine-free InstanceNorm have no gradient)
                rows.append(((torch.dot(ga, gb) / (ga.norm() * gb.norm() + 1e-30)).item(), n, ga.norm().item(), gb.norm().item()))
    assert min(rows)[0] > 0.995, sorted(rows)
    assert torch.isfinite(res[1][1]).all()


def test_tail_that_forms_the_residual_from_the_coarse_tensor_agrees_with_the_stored_residual(monkeypatch):
    """ops.TAIL_FROM_COARSE (csrc/norm.hip: tail_fwd_coarse_kernel): decoder1's residual ConvT(x) is formed inside the tail forward instead of being stored by
    upconv4_fwd and read back.  Same function; the residual enters the sum in fp32 instead of rounded to bf16: loss, reconstruction and gradients agree to bf16
    accuracy with the stored form (two samples, one ragged), and the new launch is the one that runs."""
    from nerf_mae_amd import ops
    from oracle import mae3d_oracle as O
    ora, hip = _pair(SWIN_T, torch.bfloat16, res=64, init="default")
    xs = [O.synthetic_grid((64, 64, 64), 51).cuda(), O.synthetic_grid((64, 60, 51), 52).cuda()]
    bm = O.draw_block_mask((16, 16, 16), ora.masking_prob, rng=random.Random(9))
    calls = []
    real = ops.mae_tail_fwd_from_coarse
    monkeypatch.setattr(ops, "mae_tail_fwd_from_coarse", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    res = {}
    for flag in (True, False):   # (the packer allocates the fragment buffer at the first forward: the flag is on there)
        monkeypatch.setattr(ops, "TAIL_FROM_COARSE", flag)
        n0 = len(calls)
        hip.zero_grad()
        out = hip(xs, block_mask=bm, return_pred=True)
        out[0].backward()
        torch.cuda.synchronize()
        assert len(calls) - n0 == (1 if flag else 0)
        res[int(flag)] = (out[0].item(), hip._flat_grad.clone(), out[3].clone())
    assert abs(res[0][0] - res[1][0]) <= 2e-3 * abs(res[0][0]), (res[0][0], res[1][0])
    assert_close(res[1][2], res[0][2].cpu(), 3e-2, "reconstructed grid, residual formed in the tail vs stored", elem_mult=BF16_ELEM_MULT)
    a, b = res[1][1].double(), res[0][1].double()
    assert (torch.dot(a, b) / (a.norm() * b.norm())).item() > 0.999
    rows = []
    for n, p in hip.named_parameters():
        if p.requires_grad:
            off = hip._offsets[id(p)]
            ga, gb = res[1][1][off:off + p.numel()].double(), res[0][1][off:off + p.numel()].double()
            if gb.norm() > 0:
                rows.append(((torch.dot(ga, gb) / (ga.norm() * gb.norm() + 1e-30)).item(), n, ga.norm().item(), gb.norm().item()))
    assert min(rows)[0] > 0.99, sorted(rows)[:5]
    assert torch.isfinite(res[1][1]).all()


def test_bf16_close_to_oracle_and_eval_contract():
    from oracle import mae3d_oracle as O
    res = 96
    ora, hip = _pair(TINY, torch.bfloat16, res=res, init="default")
    xs = [O.synthetic_grid((res, res, res), 21), O.synthetic_grid((res, 80, 91), 22)]
    lo, lh, l64, rows = _grads_vs_fp64(ora, hip, xs, 7)
    for a, b in zip(lh[:3], lo[:3]):
        assert abs(a.item() - b.item()) / abs(b.item()) < 1e-2
    assert_close(lh[3], lo[3], 3e-2, "reconstructed grid", elem_mult=BF16_ELEM_MULT)
    assert min(r[2] for r in rows.values()) > 0.95, min(rows.items(), key=lambda kv: kv[1][2])
    po, ph = dict(ora.named_parameters()), dict(hip.named_parameters())
    fa = torch.cat([ph[n].grad.float().cpu().flatten() for n in rows])
    fb = torch.cat([po[n].grad.flatten() for n in rows])
    assert (torch.dot(fa, fb) / (fa.norm() * fb.norm())).item() > 0.999
    # eval contract (swin_mae3d.py:1578-1594): 6-tuple, patchified shapes
    hip.eval()
    g = res // 4
    with torch.no_grad():
        out = hip([t.cuda() for t in xs], is_eval=True)
    assert len(out) == 6 and out[3].shape == (2, g, g, g, 64, 4) and out[4].shape == (2, g, g, g, 64, 1) and out[4].dtype == torch.bool
    assert out[5].shape == (2, g, g, g, 64, 4)


def test_stochastic_depth_and_train_step_decreases_loss():
    """train mode with SD noise injected identically on both sides; then a few fused AdamW steps reduce the loss."""
    from oracle import mae3d_oracle as O
    from nerf_mae_amd.trainer import FusedAdamW
    ora, hip = _pair(TINY, torch.float32, sd=0.2)
    xs = [O.synthetic_grid((32, 32, 32), 31), O.synthetic_grid((32, 32, 32), 32)]
    nblk = 8
    gen = torch.Generator().manual_seed(3)
    noise = [(torch.bernoulli(torch.full((2,), 0.8), generator=gen) / 0.8, torch.bernoulli(torch.full((2,), 0.8), generator=gen) / 0.8) for _ in range(nblk)]
    bm = O.draw_block_mask((8, 8, 8), 0.75, rng=random.Random(1))
    # oracle with the same per-block noise: patch its RowStochasticDepth modules
    it = iter([n for pair in noise for n in pair])
    for mod in ora.modules():
        if isinstance(mod, O.RowStochasticDepth):
            mod.forward = (lambda x, _it=it: x * next(_it).view(-1, 1, 1, 1, 1))
    lo = ora(xs, block_mask=bm)
    lh = hip([t.cuda() for t in xs], block_mask=bm, sd_noise=[(a.cuda(), b.cuda()) for a, b in noise])
    assert abs(lh[0].item() - lo[0].item()) / lo[0].item() < 1e-4
    opt = FusedAdamW(hip, lr=1e-3, weight_decay=1e-3, max_grad_norm=0.1)
    first = last = None
    for step in range(6):
        hip.zero_grad()
        l = hip([t.cuda() for t in xs], block_mask=bm, sd_noise=[(torch.ones(2).cuda(), torch.ones(2).cuda())] * nblk)[0]
        l.backward()
        opt.step()
        first = l.item() if first is None else first
        last = l.item()
    assert last < first


def test_nerf_rpn_encoder_contract():
    """feature_extractor.py:1155-1187: strict load, delete decoder/out/mask_token, encoder-only NCDHW features."""
    from oracle import mae3d_oracle as O
    ora, hip = _pair(TINY, torch.float32)
    del hip.decoder4, hip.decoder3, hip.decoder2, hip.decoder1, hip.out, hip.mask_token
    xb = torch.stack([O.synthetic_grid((32, 32, 32), 5), O.synthetic_grid((32, 32, 32), 6)])
    with torch.no_grad():
        fo = ora.encoder_features(xb)
        x = hip.patch_partition(xb.cuda())
        x = x + hip.pos_embed
        fh = []
        for i in range(len(hip.stages)):
            x = hip.stages[i](x)
            fh.append(torch.permute(x, [0, 4, 1, 2, 3]).contiguous())
    assert [tuple(f.shape) for f in fh] == [(2, 32, 8, 8, 8), (2, 64, 4, 4, 4), (2, 128, 2, 2, 2), (2, 256, 1, 1, 1)]
    for a, b in zip(fh, fo):
        assert relerr(a, b) < 1e-3


def test_graphed_step_recaptures_when_a_mask_keeps_more_tokens_than_the_embed_capacity():
    """trainer.GraphedTrainStep fixes the rows of the compact patch embed at capture (mean + 8 sigma of the mask distribution); a mask that keeps more tokens makes it
    re-capture with a row for every token -- forced here with a capacity of 64 rows: the replayed losses equal the eager model's on the same masks, before and
    after the overflow"""
    from nerf_mae_amd.trainer import FusedAdamW, GraphedTrainStep
    from oracle import mae3d_oracle as O
    ora, hip = _pair(SWIN_T, torch.bfloat16, res=32, init="default")
    _, ref = _pair(SWIN_T, torch.bfloat16, res=32, init="default")
    ref.load_state_dict(hip.state_dict(), strict=True)
    xs = [O.synthetic_grid((32, 32, 32), 61).cuda(), O.synthetic_grid((32, 30, 27), 62).cuda()]
    opt = FusedAdamW(hip, lr=0.0, weight_decay=0.0, max_grad_norm=0.1)     # lr 0: the weights stay put, the losses depend on the mask only
    step = GraphedTrainStep(hip, opt, 2)
    step._embed_cap = 64
    few = torch.ones(8, 8, 8, dtype=torch.uint8)
    few[:4, :4, :4] = 0          # one block kept: 64 tokens, fits
    many = torch.zeros(8, 8, 8, dtype=torch.uint8)
    many[:4, :4, :4] = 1         # seven blocks kept: 448 tokens
    for i, bm in enumerate((few, many, few)):
        losses = step(xs if i == 0 else None, bm)
        torch.cuda.synchronize()
        want = ref(xs, block_mask=bm)
        assert abs(losses[0].item() - want[0].item()) <= 2e-5 * abs(want[0].item()), (i, losses[0].item(), want[0].item())
        assert step._embed_cap == (64 if i == 0 else 512)


def test_training_trace_matches_reference_golden(golden):
    """10 optimizer steps of the whole training step (HIP forward + backward on two grids, one of them ragged; fused clip + AdamW, OneCycle
    schedule, python-random block masks) against the trace the REAL reference produced with torch.optim.AdamW / OneCycleLR /
    clip_grad_norm_ (golden g13, oracle/gen_golden_trace.py: the head_dim-32 sibling of G9).  Tolerances as for the oracle's own trace
    test: tight on the first steps, statistical afterwards (Adam turns rounding-level gradient noise into lr-sized steps; the reference
    itself drifts 2-4 % by step 10 between thread counts)."""
    from nerf_mae_amd.model import SwinTransformer_MAE3D
    from nerf_mae_amd.trainer import FusedAdamW, OneCycle
    from oracle import mae3d_oracle as O
    from oracle.gen_golden_trace import KW, STEPS
    g = golden("g13_train_trace_hd32.npz")
    kw = {k: v for k, v in KW.items() if k != "expand_dim"}
    ora = O.MAE3DOracle(pad_pos_embed=True, **kw)
    O.formula_fill_(ora)
    hip = SwinTransformer_MAE3D(compute_dtype=torch.float32, **KW)
    hip.load_state_dict(ora.state_dict(), strict=True)
    hip = hip.cuda().train()
    opt = FusedAdamW(hip, lr=1e-4, weight_decay=1e-3, max_grad_norm=0.1)
    sched = OneCycle(1e-4, STEPS)
    random.seed(13)
    trace, lrs, gn = [], [], []
    for step in range(STEPS):
        lr, b1 = sched.at(step)
        opt.set_hyper(lr=lr, beta1=b1)
        lrs.append(lr)
        hip.zero_grad()
        loss, l_rgb, l_a = hip([O.synthetic_grid((32, 32, 32), 200 + step).cuda(), O.synthetic_grid((30, 32, 27), 300 + step).cuda()])
        loss.backward()
        opt.step()
        trace.append([loss.item(), l_rgb.item(), l_a.item()])
        gn.append(opt.norm.item())
    np.testing.assert_allclose(lrs, g["lrs"], rtol=1e-6)               # OneCycle == torch OneCycleLR
    tr = np.array(trace)
    print("hip  :", tr[:, 0].round(5).tolist())
    print("ref  :", g["trace"][:, 0].round(5).tolist())
    np.testing.assert_allclose(tr[:1], g["trace"][:1], rtol=2e-4)                  # identical weights: forward parity
    np.testing.assert_allclose(np.array(gn)[:1], g["grad_norm"][:1], rtol=2e-3)    # ... and the pre-clip global gradient norm (6.7e4)
    # From the first update on the comparison is statistical: Adam's first step moves every parameter by +-lr whatever the size of its
    # gradient, so the elements whose gradient is rounding noise (1e-4 relative between the two fp32 paths) go opposite ways, and with a
    # pre-clip gradient norm of 6.7e4 the trajectory is chaotic (the reference itself drifts 2-4 % between thread counts).  The curves
    # must track each other and end at the same level.
    # (run to run, through the fp32-atomic summation order alone, single steps of this trace move by +-6 %)
    np.testing.assert_allclose(tr[:3, 0], g["trace"][:3, 0], rtol=2e-2)
    np.testing.assert_allclose(tr[:, 0], g["trace"][:, 0], rtol=0.2)
    np.testing.assert_allclose(tr[:, 1:], g["trace"][:, 1:], rtol=0.35)
    assert abs(tr[-3:, 0].mean() - g["trace"][-3:, 0].mean()) < 0.08 * g["trace"][-3:, 0].mean()
    assert tr[-1, 0] < 0.45 * tr[0, 0]


@pytest.mark.parametrize("dtype,tol,use_graph", [(torch.float32, 2e-2, False), (torch.bfloat16, 5e-2, False), (torch.bfloat16, 5e-2, True)],
                         ids=["fp32", "bf16", "bf16-hipgraph"])
def test_well_conditioned_training_trace_within_2_percent_every_step(golden, dtype, tol, use_graph):
    """recon-loss curve against the REAL reference at every step (golden g14, oracle/gen_golden_trace2.py: reference init distributions,
    torch AdamW(eps 1e-3) + OneCycleLR + clip_grad_norm_, ten steps on two fixed grids with python-random block masks; reproducible to
    0.35 % between 1 and 8 CPU threads): fp32 within 2 %, bf16 within 5 % -- loss, loss_rgb, loss_alpha and the pre-clip gradient norm.
    A wrong beta1 cycle, bias correction, weight decay or schedule step moves the curve by more than that from step 2 on (checked by
    breaking each of them).  The graph variant runs the same steps through GraphedTrainStep (the benched path)."""
    from nerf_mae_amd.model import SwinTransformer_MAE3D, draw_block_mask
    from nerf_mae_amd.trainer import FusedAdamW, GraphedTrainStep, OneCycle
    from oracle import mae3d_oracle as O
    from oracle.gen_golden_trace2 import CLIP, EPS, KW, LR, SEED, STEPS, WD, grids
    g = golden("g14_train_trace_wellcond.npz")
    hip = SwinTransformer_MAE3D(compute_dtype=dtype, **KW)
    O.seeded_reference_init_(hip, SEED)     # per-parameter generators: bit-identical to the weights the reference trained from
    hip = hip.cuda().train()
    opt = FusedAdamW(hip, lr=LR, weight_decay=WD, max_grad_norm=CLIP, eps=EPS)
    sched = OneCycle(LR, STEPS)
    random.seed(14)
    xs = [t.cuda() for t in grids()]
    step_fn = GraphedTrainStep(hip, opt, 2) if use_graph else None
    trace, gn = [], []
    for step in range(STEPS):
        lr, b1 = sched.at(step)
        opt.set_hyper(lr=lr, beta1=b1)
        if use_graph:
            losses = step_fn(xs if step == 0 else None, draw_block_mask((8, 8, 8), 0.75, rng=random))
            trace.append([float(v) for v in losses])
        else:
            hip.zero_grad()
            loss, l_rgb, l_a = hip(xs)
            loss.backward()
            opt.step()
            trace.append([loss.item(), l_rgb.item(), l_a.item()])
        gn.append(opt.norm.item())
    tr = np.array(trace)
    print("hip:", tr[:, 0].round(5).tolist())
    print("ref:", g["trace"][:, 0].round(5).tolist())
    print("max rel err: trace", float(np.abs(tr / g["trace"] - 1).max()), "per column", np.abs(tr / g["trace"] - 1).max(0).round(5).tolist(),
          "grad norm", float(np.abs(np.array(gn) / g["grad_norm"] - 1).max()))
    np.testing.assert_allclose(tr, g["trace"], rtol=tol)
    print("grad-norm rel err per step", np.abs(np.array(gn) / g["grad_norm"] - 1).round(4).tolist())
    # the pre-clip gradient norm: tight while the two runs still hold (nearly) the same weights; from step 4 on it is the most sensitive
    # quantity of the run (fp32 repeats of THIS implementation differ by 1-5 % there through the summation order of the fp32 atomics, with the
    # loss curve still within 0.6 %), so the late steps only bound it
    np.testing.assert_allclose(np.array(gn)[:4], g["grad_norm"][:4], rtol=2 * tol)
    np.testing.assert_allclose(np.array(gn)[4:], g["grad_norm"][4:], rtol=max(2 * tol, 0.12))
