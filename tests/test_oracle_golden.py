"""CPU: pins oracle/mae3d_oracle.py (the checker) against golden vectors produced by the REAL
reference (oracle/gen_golden.py, run in the build container).  No GPU, no /root/reference."""
import random

import numpy as np
import pytest
import torch

from oracle import mae3d_oracle as O

TOL = dict(rtol=2e-5, atol=2e-5)


def close(a, b, rtol=2e-5, atol=2e-5):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("tag,shape,shift", [("a_8_s0", (2, 8, 8, 8), 0), ("b_8_s2", (2, 8, 8, 8), 2),
                                             ("c_5_s2", (1, 5, 5, 5), 2), ("d_10_s2", (1, 10, 10, 10), 2),
                                             ("e_2_s2", (1, 2, 2, 2), 2), ("f_684_s2", (1, 6, 8, 4), 2)])
def test_g1_window_attention(golden, tag, shape, shift):
    g = golden("g1_window_attention.npz")
    C, heads = 24, 3
    att = O.WindowAttention3D(C, heads, [shift] * 3)
    with torch.no_grad():
        for n, p in att.named_parameters():
            sc = 0.5 if "table" in n else (0.05 if n.endswith("bias") else 0.25)
            p.copy_(O.formula_tensor("g1." + n, p.shape, sc))
    x = O.formula_tensor("g1.x." + tag, shape + (C,), 1.0).requires_grad_(True)
    y = att(x)
    (y * O.formula_tensor("g1.dy." + tag, y.shape, 1.0)).sum().backward()
    close(y, g[tag + ".y"])
    close(x.grad, g[tag + ".dx"], 1e-4, 1e-4)
    for n, p in att.named_parameters():
        close(p.grad, g[tag + ".d_" + n], 1e-4, 2e-4)


def test_g2_tables(golden):
    g = golden("g2_tables.npz")
    assert np.array_equal(O.rel_pos_index(4).numpy().astype(np.int16), g["rel_index"])
    close(O.sincos_pos_embed_3d(24, 8).astype(np.float32), g["pos_embed_24_8"], 1e-6, 1e-6)
    big = O.sincos_pos_embed_3d(96, 40).astype(np.float32)
    close(big[0, ::13, ::11, ::7, :], g["pos_embed_96_40_samples"], 1e-6, 1e-6)
    s = np.array([big.astype(np.float64).sum(), np.abs(big.astype(np.float64)).sum()])
    np.testing.assert_allclose(s, g["pos_embed_96_40_sum"], rtol=1e-9)


@pytest.mark.parametrize("tag,shape", [("even", (2, 8, 8, 8)), ("odd", (1, 5, 5, 5)), ("mixed", (1, 6, 5, 4))])
def test_g3_patch_merging(golden, tag, shape):
    g = golden("g3_patch_merging.npz")
    pm = O.PatchMerging3D(8)
    with torch.no_grad():
        pm.reduction.weight.copy_(O.formula_tensor("g3.red", pm.reduction.weight.shape, 0.2))
        pm.norm.weight.copy_(O.formula_tensor("g3.nw", pm.norm.weight.shape, 0.2, 1.0))
        pm.norm.bias.copy_(O.formula_tensor("g3.nb", pm.norm.bias.shape, 0.1))
    x = O.formula_tensor("g3.x." + tag, shape + (8,), 1.0).requires_grad_(True)
    y = pm(x)
    (y * O.formula_tensor("g3.dy." + tag, y.shape, 1.0)).sum().backward()
    close(y, g[tag + ".y"])
    close(x.grad, g[tag + ".dx"], 1e-4, 1e-4)
    close(pm.reduction.weight.grad, g[tag + ".d_red"], 1e-4, 1e-4)
    close(pm.norm.weight.grad, g[tag + ".d_nw"], 1e-4, 1e-4)
    close(pm.norm.bias.grad, g[tag + ".d_nb"], 1e-4, 1e-4)


@pytest.mark.parametrize("tag,cin,cout,k,skip,sp", [("k2_skip", 16, 8, 2, True, 3), ("k4_noskip", 16, 8, 4, False, 2)])
def test_g4_up_block(golden, tag, cin, cout, k, skip, sp):
    g = golden("g4_up_block.npz")
    blk = O.UpBlock3D(cin, cout, k, use_skip=skip)
    with torch.no_grad():
        for n, p in blk.named_parameters():
            p.copy_(O.formula_tensor("g4." + tag + n, p.shape, 0.05 if n.endswith("bias") else 0.15))
    x = O.formula_tensor("g4.x." + tag, (2, cin, sp, sp, sp), 1.0).requires_grad_(True)
    s = O.formula_tensor("g4.s." + tag, (2, cout, sp * k, sp * k, sp * k), 1.0).requires_grad_(True) if skip else None
    y = blk(x, s)
    (y * O.formula_tensor("g4.dy." + tag, y.shape, 1.0)).sum().backward()
    close(y, g[tag + ".y"], 1e-4, 1e-4)
    close(x.grad, g[tag + ".dx"], 1e-3, 1e-4)
    if skip:
        close(s.grad, g[tag + ".ds"], 1e-3, 1e-4)
    for n, p in blk.named_parameters():
        # conv biases feeding an InstanceNorm have mathematically-zero grads (rounding noise in both)
        close(p.grad, g[tag + ".d_" + n], 1e-3, 2e-4)


def test_g5_masking_rng(golden):
    g = golden("g5_masks.npz")
    for key, gsz, seed, p in [("blocks_40_seed1234", 40, 1234, 0.75), ("blocks_8_seed1234", 8, 1234, 0.75),
                              ("blocks_40_seed7_p50", 40, 7, 0.5)]:
        random.seed(seed)
        m = O.draw_block_mask((gsz,) * 3, p)[::4, ::4, ::4].numpy().astype(np.uint8)
        assert np.array_equal(m, g[key]), key
    # SURVEY 8(c)(iii) known answer
    random.seed(1234)
    bits = [random.random() < 0.75 for _ in range(1000)]
    assert sum(bits) == 743 and "".join("1" if b else "0" for b in bits[:16]) == "0110011101101111"
    assert int(g["blocks_40_seed1234"].sum()) == 743


def test_g6_loss(golden):
    g = golden("g6_loss.npz")
    x6 = torch.stack([O.synthetic_grid((32, 32, 32), 3), O.synthetic_grid((32, 32, 32), 4)])
    valid = torch.ones_like(x6)
    valid[1, :, 28:] = 0
    valid[1, :, :, 24:] = 0
    x6 = x6 * valid
    pred = O.formula_tensor("g6.pred", x6.shape, 1.5).requires_grad_(True)
    tm = torch.from_numpy(g["token_mask"])
    l, lr, la, *_ = O.mae_loss(x6, pred, valid, tm)
    l.backward()
    np.testing.assert_allclose([l.item(), lr.item(), la.item()], g["losses"], rtol=1e-5)
    close(pred.grad[:, :, ::5, ::7, ::3], g["dpred_samples"], 1e-4, 1e-9)
    np.testing.assert_allclose([pred.grad.double().sum().item(), pred.grad.double().abs().sum().item()], g["dpred_sum"], rtol=1e-4)


KW7 = dict(resolution=32, masking_prob=0.75)


def test_g7_survey_known_answer(golden):
    """seeded fresh-init swin_t@32 -> loss 1.7251027822 (same RNG consumption order
    as the reference: torch init draws, python-random mask draws, torch SD draws)."""
    g = golden("g7_swin_t_32.npz")
    torch.manual_seed(0)
    random.seed(0)
    m = O.build_oracle("swin_t", stochastic_depth_prob=0.1, **KW7)
    assert sum(p.numel() for p in m.parameters()) == int(g["nparam"][0]) == 48747130
    # NB: parameter *creation order* differs from the reference, so fresh-init weights differ; the
    # known answer is therefore checked through a state_dict transplant in the reference-side
    # generator, and here only structurally (key set + shapes) -- see test_state_dict_keys.
    out = m([torch.rand(4, 32, 32, 32)])
    assert all(torch.isfinite(v) for v in out)
    np.testing.assert_allclose(g["survey_known_answer"], [1.7251027822, 1.6379609108, 0.0871418193], rtol=1e-8)


def test_g7_formula_model(golden):
    g = golden("g7_swin_t_32.npz")
    m = O.build_oracle("swin_t", stochastic_depth_prob=0.0, **KW7)
    O.formula_fill_(m)
    xg = [O.synthetic_grid((32, 32, 32), 11)]
    random.seed(42)
    l, lr, la, pred = m(xg, return_pred=True)
    l.backward()
    np.testing.assert_allclose([l.item(), lr.item(), la.item()], g["losses"], rtol=2e-5)
    close(pred, g["pred"], 2e-4, 2e-4)
    gn = {n: float(p.grad.double().norm()) for n, p in m.named_parameters() if p.grad is not None}
    for n, v in zip(g["grad_names"], g["grad_norms"]):
        n = str(n)
        if n.endswith("conv1.bias") or n.endswith("conv2.bias") or n.endswith("conv3.bias"):
            assert gn[n] < 1e-4 and v < 1e-4  # cancelled by InstanceNorm: rounding noise on both sides
            continue
        np.testing.assert_allclose(gn[n], v, rtol=2e-3, atol=1e-7, err_msg=n)
    close(m.mask_token.grad, g["d_mask_token"], 1e-3, 1e-6)
    close(m.out.conv.weight.grad, g["d_out_w"], 1e-3, 1e-6)
    close(m.stages[0][1].attn.relative_position_bias_table.grad, g["d_bias_table_s0b1"], 1e-3, 1e-6)


def test_g7_stochastic_depth_seeded(golden):
    g = golden("g7_swin_t_32.npz")
    m = O.build_oracle("swin_t", stochastic_depth_prob=0.1, **KW7)
    O.formula_fill_(m)
    random.seed(43)
    torch.manual_seed(43)
    out = m([O.synthetic_grid((32, 32, 32), 11), O.synthetic_grid((32, 32, 32), 12)])
    np.testing.assert_allclose([float(v) for v in out], g["losses_sd_seed43"], rtol=2e-5)


KW8 = dict(embed_dim=24, depths=[2, 2, 2, 2], num_heads=[3, 6, 12, 24], stochastic_depth_prob=0.0, resolution=32, masking_prob=0.75)


def test_g8_tiny_model_variable_inputs(golden):
    g = golden("g8_tiny_model.npz")
    m = O.MAE3DOracle(**KW8)
    O.formula_fill_(m)
    x8 = [O.synthetic_grid((30, 28, 32), 21), O.synthetic_grid((32, 32, 20), 22)]
    random.seed(8)
    l, lr, la, pred = m(x8, return_pred=True)
    l.backward()
    np.testing.assert_allclose([l.item(), lr.item(), la.item()], g["losses"], rtol=2e-5)
    close(pred, g["pred"], 2e-4, 2e-4)
    params = dict(m.named_parameters())
    for n, v in zip(g["grad_names"], g["grad_norms"]):
        n = str(n)
        if n.endswith(("conv1.bias", "conv2.bias", "conv3.bias")):
            continue
        np.testing.assert_allclose(float(params[n].grad.double().norm()), v, rtol=2e-3, atol=1e-7, err_msg=n)
    for k in g.files:
        if k.startswith("d_"):
            close(params[k[2:]].grad, g[k], 2e-3, 2e-6)
    random.seed(8)
    ev = m(x8, is_eval=True)
    assert [list(t.shape) for t in ev[3:]] == g["eval_shapes"].tolist()
    assert int(ev[4].sum()) == int(g["eval_mask_count"][0])
    np.testing.assert_allclose([ev[3].double().sum().item(), ev[3].double().abs().sum().item()], g["eval_pred_sum"], rtol=1e-4)
    # G10: nerf_rpn encoder-feature contract
    xb = torch.cat([O.pad_grid(t, 32)[0] for t in x8], 0)
    with torch.no_grad():
        feats = m.encoder_features(xb)
    for i, f in enumerate(feats):
        assert list(f.shape) == g[f"feat{i}_shape"].tolist()
        np.testing.assert_allclose([f.double().sum().item(), f.double().abs().sum().item()], g[f"feat{i}_sum"], rtol=1e-4, atol=1e-3)
        if i >= 2:
            close(f, g[f"feat{i}"], 2e-4, 2e-4)


def test_g9_training_trace(golden):
    g = golden("g9_train_trace.npz")
    m = O.MAE3DOracle(**KW8)
    O.formula_fill_(m)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-3)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-4, total_steps=10)
    random.seed(9)
    trace = []
    for step in range(10):
        opt.zero_grad()
        loss, lr_, la_ = m([O.synthetic_grid((32, 32, 32), 100 + step)])
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 0.1)
        opt.step()
        sch.step()
        trace.append([float(loss), float(lr_), float(la_)])
    # Adam turns rounding-level gradient noise into lr-sized steps, so the trace is chaotic: the
    # reference itself drifts 2-4 % by step 10 between 1 and 8 CPU threads ("loss-curve
    # tolerance").  Tight on the first steps, statistical afterwards.
    np.testing.assert_allclose(np.array(trace)[:3], g["trace"][:3], rtol=2e-4)
    np.testing.assert_allclose(np.array(trace), g["trace"], rtol=8e-2)


def test_g13_training_trace_head_dim_32(golden):
    """the oracle against the reference's own 10-step trace at head_dim 32 on two grids (oracle/gen_golden_trace.py)"""
    from oracle.gen_golden_trace import KW, STEPS
    g = golden("g13_train_trace_hd32.npz")
    torch.set_num_threads(8)
    m = O.MAE3DOracle(pad_pos_embed=True, **{k: v for k, v in KW.items() if k != "expand_dim"})
    O.formula_fill_(m)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-3)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-4, total_steps=STEPS)
    random.seed(13)
    trace = []
    for step in range(STEPS):
        opt.zero_grad()
        loss, lr_, la_ = m([O.synthetic_grid((32, 32, 32), 200 + step), O.synthetic_grid((30, 32, 27), 300 + step)])
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 0.1)
        opt.step()
        sch.step()
        trace.append([float(loss), float(lr_), float(la_)])
    np.testing.assert_allclose(np.array(trace)[:3], g["trace"][:3], rtol=2e-4)
    np.testing.assert_allclose(np.array(trace), g["trace"], rtol=8e-2)


def test_g14_well_conditioned_training_trace(golden):
    """the oracle against the reference's well-conditioned 10-step trace (oracle/gen_golden_trace2.py: reference init distributions seeded
    per parameter, AdamW eps 1e-3, OneCycle, clip 0.1): every step within 1 % -- the reference itself moves 0.35 % between 1 and 8 threads"""
    from oracle.gen_golden_trace2 import CLIP, EPS, KW, LR, SEED, STEPS, WD, grids
    g = golden("g14_train_trace_wellcond.npz")
    assert float(g["thread_deviation"]) < 1e-2
    torch.set_num_threads(8)
    m = O.MAE3DOracle(pad_pos_embed=True, **{k: v for k, v in KW.items() if k != "expand_dim"})
    O.seeded_reference_init_(m, SEED)
    opt = torch.optim.AdamW(m.parameters(), lr=LR, weight_decay=WD, eps=EPS)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=LR, total_steps=STEPS)
    random.seed(14)
    xs, trace, gn = grids(), [], []
    for step in range(STEPS):
        opt.zero_grad()
        loss, lr_, la_ = m(xs)
        loss.backward()
        gn.append(float(torch.nn.utils.clip_grad_norm_(m.parameters(), CLIP)))
        opt.step()
        sch.step()
        trace.append([loss.item(), lr_.item(), la_.item()])
    np.testing.assert_allclose(np.array(trace)[:2], g["trace"][:2], rtol=2e-4)
    np.testing.assert_allclose(np.array(trace), g["trace"], rtol=1e-2)
    np.testing.assert_allclose(np.array(gn), g["grad_norm"], rtol=3e-2)


def test_state_dict_keys_match_reference_contract():
    """key set / shapes / counts (215 keys swin_t, 383 swin_s)."""
    for name, nkeys in [("swin_t", 215), ("swin_s", 383)]:
        m = O.build_oracle(name, resolution=32)
        sd = m.state_dict()
        assert len(sd) == nkeys
        assert sd["pos_embed"].shape == (1, 8, 8, 8, 96)
        assert sd["stages.1.0.reduction.weight"].shape == (192, 768)
        assert sd["stages.2.1.attn.relative_position_index"].dtype == torch.int64
        assert sd["decoder1.transp_conv.weight"].shape == (96, 48, 4, 4, 4)
        assert sd["stages.0.0.mlp.3.weight"].shape == (96, 384)
        assert "decoder1.conv_block.conv3.weight" not in sd and "decoder2.conv_block.conv3.weight" in sd
    m = O.build_oracle("swin_s", resolution=160)
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 70040938
