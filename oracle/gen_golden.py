"""Generate tests/golden/*.npz by running the REAL reference (read-only, /root/reference)
in the build container.  TEST INFRASTRUCTURE ONLY -- never imported by the product.

Run:  python oracle/gen_golden.py            (needs /root/reference; not available on the GPU box)

The reference cannot be imported unmodified here: it needs torchvision 0.13.1 (absent) and
numpy<1.24 (`np.float`).  Two shims are injected *in this process only* (SURVEY 8(c)):
  1. fake modules torchvision.ops.stochastic_depth.StochasticDepth / torchvision.ops.misc.{MLP,Permute}
     restating the published torchvision 0.13.1 semantics,
  2. np.float = float.
Weights/inputs are closed-form ("formula-filled", oracle.mae3d_oracle.formula_tensor) so only
inputs' seeds and the reference OUTPUTS are stored.
"""
from __future__ import annotations

import os
import random
import sys
import types

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.mae3d_oracle import formula_fill_, formula_tensor, synthetic_grid  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def _install_shims():
    np.float = float  # torch_utils.py:42

    class StochasticDepth(nn.Module):
        def __init__(self, p, mode):
            super().__init__()
            self.p, self.mode = p, mode

        def forward(self, x):
            if not self.training or self.p == 0.0:
                return x
            s = 1.0 - self.p
            size = [x.shape[0]] + [1] * (x.ndim - 1) if self.mode == "row" else [1] * x.ndim
            noise = torch.empty(size, dtype=x.dtype, device=x.device).bernoulli_(s)
            if s > 0.0:
                noise.div_(s)
            return x * noise

    class MLP(nn.Sequential):
        def __init__(self, in_channels, hidden_channels, norm_layer=None, activation_layer=nn.ReLU,
                     inplace=True, bias=True, dropout=0.0):
            params = {} if inplace is None else {"inplace": inplace}
            layers, d = [], in_channels
            for h in hidden_channels[:-1]:
                layers += [nn.Linear(d, h, bias=bias)]
                if norm_layer is not None:
                    layers += [norm_layer(h)]
                layers += [activation_layer(**params), nn.Dropout(dropout, **params)]
                d = h
            layers += [nn.Linear(d, hidden_channels[-1], bias=bias), nn.Dropout(dropout, **params)]
            super().__init__(*layers)

    class Permute(nn.Module):
        def __init__(self, dims):
            super().__init__()
            self.dims = dims

        def forward(self, x):
            return torch.permute(x, self.dims)

    tv = types.ModuleType("torchvision")
    ops = types.ModuleType("torchvision.ops")
    sd = types.ModuleType("torchvision.ops.stochastic_depth")
    misc = types.ModuleType("torchvision.ops.misc")
    sd.StochasticDepth, misc.MLP, misc.Permute = StochasticDepth, MLP, Permute
    tv.ops, ops.stochastic_depth, ops.misc = ops, sd, misc
    sys.modules.update({"torchvision": tv, "torchvision.ops": ops,
                        "torchvision.ops.stochastic_depth": sd, "torchvision.ops.misc": misc})


def _np(t):
    return t.detach().cpu().numpy()


def main():
    assert os.path.isdir(REF), "reference not mounted"
    _install_shims()
    sys.path.insert(0, REF)
    from nerf_mae.model.mae import swin_mae3d as R
    from nerf_mae.model.mae import torch_utils as RU
    from nerf_mae.model.mae import unetr_block as RB

    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)

    # ---- G1: shifted_window_attention fwd + grads ------------------------------------------
    g1 = {}
    C, heads = 24, 3
    for tag, shape, shift in [("a_8_s0", (2, 8, 8, 8), 0), ("b_8_s2", (2, 8, 8, 8), 2),
                              ("c_5_s2", (1, 5, 5, 5), 2), ("d_10_s2", (1, 10, 10, 10), 2),
                              ("e_2_s2", (1, 2, 2, 2), 2), ("f_684_s2", (1, 6, 8, 4), 2)]:
        att = R.ShiftedWindowAttention(C, [4, 4, 4], [shift] * 3, heads)
        with torch.no_grad():
            for n, p in att.named_parameters():
                sc = 0.5 if "table" in n else (0.05 if n.endswith("bias") else 0.25)
                p.copy_(formula_tensor("g1." + n, p.shape, sc))
        x = formula_tensor("g1.x." + tag, shape + (C,), 1.0).requires_grad_(True)
        y = att(x)
        w = formula_tensor("g1.dy." + tag, y.shape, 1.0)
        (y * w).sum().backward()
        g1[tag + ".y"] = _np(y)
        g1[tag + ".dx"] = _np(x.grad)
        for n, p in att.named_parameters():
            g1[tag + ".d_" + n] = _np(p.grad)
    np.savez_compressed(os.path.join(OUT, "g1_window_attention.npz"), **g1)

    # ---- G2: fixed tables ------------------------------------------------------------------
    att = R.ShiftedWindowAttention(24, [4, 4, 4], [0, 0, 0], 3)
    pe_small = RU.get_3d_sincos_pos_embed(24, 8).astype(np.float32)
    pe_big = RU.get_3d_sincos_pos_embed(96, 40).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "g2_tables.npz"),
                        rel_index=_np(att.relative_position_index).astype(np.int16),
                        pos_embed_24_8=pe_small,
                        pos_embed_96_40_sum=np.array([pe_big.astype(np.float64).sum(),
                                                      np.abs(pe_big.astype(np.float64)).sum()]),
                        pos_embed_96_40_samples=pe_big[0, ::13, ::11, ::7, :].copy())

    # ---- G3: PatchMerging ------------------------------------------------------------------
    g3 = {}
    for tag, shape in [("even", (2, 8, 8, 8)), ("odd", (1, 5, 5, 5)), ("mixed", (1, 6, 5, 4))]:
        pm = R.PatchMerging(8)
        with torch.no_grad():
            pm.reduction.weight.copy_(formula_tensor("g3.red", pm.reduction.weight.shape, 0.2))
            pm.norm.weight.copy_(formula_tensor("g3.nw", pm.norm.weight.shape, 0.2, 1.0))
            pm.norm.bias.copy_(formula_tensor("g3.nb", pm.norm.bias.shape, 0.1))
        x = formula_tensor("g3.x." + tag, shape + (8,), 1.0).requires_grad_(True)
        y = pm(x)
        (y * formula_tensor("g3.dy." + tag, y.shape, 1.0)).sum().backward()
        g3[tag + ".y"], g3[tag + ".dx"] = _np(y), _np(x.grad)
        g3[tag + ".d_red"], g3[tag + ".d_nw"], g3[tag + ".d_nb"] = _np(pm.reduction.weight.grad), _np(pm.norm.weight.grad), _np(pm.norm.bias.grad)
    np.savez_compressed(os.path.join(OUT, "g3_patch_merging.npz"), **g3)

    # ---- G4: UnetrUpBlock ------------------------------------------------------------------
    g4 = {}
    for tag, cin, cout, k, skip, sp in [("k2_skip", 16, 8, 2, True, 3), ("k4_noskip", 16, 8, 4, False, 2)]:
        blk = RB.UnetrUpBlock(cin, cout, 3, k, res_block=True, use_skip=skip)
        with torch.no_grad():
            for n, p in blk.named_parameters():
                p.copy_(formula_tensor("g4." + tag + n, p.shape, 0.05 if n.endswith("bias") else 0.15))
        x = formula_tensor("g4.x." + tag, (2, cin, sp, sp, sp), 1.0).requires_grad_(True)
        s = formula_tensor("g4.s." + tag, (2, cout, sp * k, sp * k, sp * k), 1.0).requires_grad_(True) if skip else None
        y = blk(x, s)
        (y * formula_tensor("g4.dy." + tag, y.shape, 1.0)).sum().backward()
        g4[tag + ".y"], g4[tag + ".dx"] = _np(y), _np(x.grad)
        if skip:
            g4[tag + ".ds"] = _np(s.grad)
        for n, p in blk.named_parameters():
            g4[tag + ".d_" + n] = _np(p.grad)
    np.savez_compressed(os.path.join(OUT, "g4_up_block.npz"), **g4)

    # ---- G5: masking RNG -------------------------------------------------------------------
    def ref_mask(model, g, seed, p):
        random.seed(seed)
        _, m = model.window_masking_3d(torch.zeros(1, g, g, g, 4), p_remove=p, mask_token=torch.zeros(4))
        return _np(m)[0, ::4, ::4, ::4, 0].astype(np.uint8)

    tiny = R.SwinTransformer_MAE3D_New(patch_size=[4] * 3, embed_dim=24, depths=[2, 2, 2, 2], num_heads=[3, 6, 12, 24],
                                       window_size=[4] * 3, resolution=32, masking_prob=0.75)
    np.savez_compressed(os.path.join(OUT, "g5_masks.npz"),
                        blocks_40_seed1234=ref_mask(tiny, 40, 1234, 0.75), blocks_8_seed1234=ref_mask(tiny, 8, 1234, 0.75),
                        blocks_40_seed7_p50=ref_mask(tiny, 40, 7, 0.5))

    # ---- G6: forward_loss on hand-built tensors -----------------------------------------------
    x6 = torch.stack([synthetic_grid((32, 32, 32), 3), synthetic_grid((32, 32, 32), 4)])
    valid = torch.ones_like(x6)
    valid[1, :, 28:] = 0
    valid[1, :, :, 24:] = 0
    x6 = x6 * valid
    pred6 = formula_tensor("g6.pred", x6.shape, 1.5).requires_grad_(True)
    random.seed(5)
    tm = torch.zeros(2, 8, 8, 8, 1)
    for a in (0, 4):
        for b in (0, 4):
            for c in (0, 4):
                if random.random() < 0.6:
                    tm[:, a:a + 4, b:b + 4, c:c + 4] = 1
    l, lr, la = tiny.forward_loss(x6, pred6, valid, tm)
    l.backward()
    np.savez_compressed(os.path.join(OUT, "g6_loss.npz"), token_mask=_np(tm), losses=np.array([l.item(), lr.item(), la.item()], np.float64),
                        dpred_sum=np.array([pred6.grad.double().sum().item(), pred6.grad.double().abs().sum().item()]),
                        dpred_samples=_np(pred6.grad)[:, :, ::5, ::7, ::3].copy())

    # ---- G7: whole model config 1 (swin_t, 32^3) ------------------------------------------
    kw = dict(patch_size=[4] * 3, window_size=[4] * 3, expand_dim=True, resolution=32, masking_prob=0.75,
              embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24])
    torch.manual_seed(0)
    random.seed(0)
    m = R.SwinTransformer_MAE3D_New(stochastic_depth_prob=0.1, **kw)
    nparam = sum(p.numel() for p in m.parameters())
    xr = [torch.rand(4, 32, 32, 32)]
    lo = [float(v) for v in m(xr)]
    # formula-filled, SD off, seeded mask
    m2 = R.SwinTransformer_MAE3D_New(stochastic_depth_prob=0.0, **kw)
    formula_fill_(m2)
    xg = [synthetic_grid((32, 32, 32), 11)]
    random.seed(42)
    l7 = m2(xg)
    l7[0].backward()
    gn = {n: float(p.grad.double().norm()) for n, p in m2.named_parameters() if p.grad is not None}
    random.seed(42)
    with torch.no_grad():
        xb, vb = m2.transform(xg)
        pred7, _ = m2.forward_encoder_ecoder(torch.cat(xb, 0))
    # train-mode SD with torch seed
    m3 = R.SwinTransformer_MAE3D_New(stochastic_depth_prob=0.1, **kw)
    formula_fill_(m3)
    random.seed(43)
    torch.manual_seed(43)
    l7sd = [float(v) for v in m3([synthetic_grid((32, 32, 32), 11), synthetic_grid((32, 32, 32), 12)])]
    np.savez_compressed(os.path.join(OUT, "g7_swin_t_32.npz"),
                        survey_known_answer=np.array(lo, np.float64), nparam=np.array([nparam]),
                        losses=np.array([float(v) for v in l7], np.float64), pred=_np(pred7),
                        grad_names=np.array(list(gn.keys())), grad_norms=np.array(list(gn.values()), np.float64),
                        d_mask_token=_np(m2.mask_token.grad), d_out_w=_np(m2.out.conv.weight.grad),
                        d_bias_table_s0b1=_np(m2.stages[0][1].attn.relative_position_bias_table.grad),
                        losses_sd_seed43=np.array(l7sd, np.float64))

    # ---- G8: tiny-width whole model, variable-size input ------------------------------------
    kw8 = dict(patch_size=[4] * 3, window_size=[4] * 3, expand_dim=True, resolution=32, masking_prob=0.75,
               embed_dim=24, depths=[2, 2, 2, 2], num_heads=[3, 6, 12, 24], stochastic_depth_prob=0.0)
    m8 = R.SwinTransformer_MAE3D_New(**kw8)
    formula_fill_(m8)
    x8 = [synthetic_grid((30, 28, 32), 21), synthetic_grid((32, 32, 20), 22)]
    random.seed(8)
    l8 = m8(x8)
    l8[0].backward()
    g8 = {"losses": np.array([float(v) for v in l8], np.float64)}
    names = [n for n, p in m8.named_parameters() if p.grad is not None]
    g8["grad_names"] = np.array(names)
    g8["grad_norms"] = np.array([float(dict(m8.named_parameters())[n].grad.double().norm()) for n in names])
    for n in ["mask_token", "patch_partition.0.weight", "stages.1.0.reduction.weight", "decoder1.transp_conv.weight",
              "decoder1.conv_block.conv1.weight", "decoder4.conv_block.conv3.weight", "stages.3.2.attn.qkv.weight",
              "stages.0.1.attn.relative_position_bias_table", "out.conv.bias"]:
        g8["d_" + n] = _np(dict(m8.named_parameters())[n].grad)
    random.seed(8)
    ev = m8(x8, is_eval=True)
    g8["eval_pred_sum"] = np.array([ev[3].double().sum().item(), ev[3].double().abs().sum().item()])
    g8["eval_mask_count"] = np.array([int(ev[4].sum())])
    g8["eval_shapes"] = np.array([list(ev[3].shape), list(ev[4].shape), list(ev[5].shape)])
    random.seed(8)
    with torch.no_grad():
        xb, vb = m8.transform(x8)
        pred8, _ = m8.forward_encoder_ecoder(torch.cat(xb, 0))
    g8["pred"] = _np(pred8)
    # G10: encoder-only features as nerf_rpn consumes them (feature_extractor.py:1176-1187)
    with torch.no_grad():
        t = m8.patch_partition(torch.cat(xb, 0))
        t = t + m8.pos_embed
        for i, st in enumerate(m8.stages):
            t = st(t)
            f = torch.permute(t, [0, 4, 1, 2, 3]).contiguous()
            g8[f"feat{i}_shape"] = np.array(f.shape)
            g8[f"feat{i}_sum"] = np.array([f.double().sum().item(), f.double().abs().sum().item()])
            if i >= 2:
                g8[f"feat{i}"] = _np(f)
    np.savez_compressed(os.path.join(OUT, "g8_tiny_model.npz"), **g8)

    # ---- G9: 10-step training trace (AdamW + clip 0.1 + OneCycleLR), run_swin_mae3d.py:588-598,644-669
    m9 = R.SwinTransformer_MAE3D_New(**kw8)
    formula_fill_(m9)
    opt = torch.optim.AdamW(m9.parameters(), lr=1e-4, weight_decay=1e-3)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-4, total_steps=10)
    random.seed(9)
    trace, lrs, betas = [], [], []
    for step in range(10):
        opt.zero_grad()
        loss, lr_, la_ = m9([synthetic_grid((32, 32, 32), 100 + step)])
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m9.parameters(), 0.1)
        lrs.append(opt.param_groups[0]["lr"])
        betas.append(opt.param_groups[0]["betas"][0])
        opt.step()
        sch.step()
        trace.append([float(loss), float(lr_), float(la_)])
    fin = {n: [p.double().sum().item(), p.double().abs().sum().item()] for n, p in m9.named_parameters()}
    np.savez_compressed(os.path.join(OUT, "g9_train_trace.npz"), trace=np.array(trace, np.float64), lrs=np.array(lrs), beta1=np.array(betas),
                        final_names=np.array(list(fin.keys())), final_sums=np.array(list(fin.values())))
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden written to", OUT, "total bytes", tot, "G7 known answer", lo, "nparam", nparam)


if __name__ == "__main__":
    main()
