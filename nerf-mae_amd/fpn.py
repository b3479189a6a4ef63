"""nerf_rpn backbone drop-in: host-side mirror of `FPN` (nerf_rpn/model/fpn.py:8-185, the configuration the
Swin backbones use: start_level 0, no extra levels, nearest top-down) and of `SwinTransformer_FPN_Pretrained_Skip`
(nerf_rpn/model/feature_extractor.py:1067-1187): MAE-pretrained encoder, decoders/head/mask token deleted, FPN neck on the four
stage outputs, `forward(x (B,4,R,R,R)) -> tuple of 4 NCDHW maps`.  Same constructor arguments, attribute names (`base`,
`fpn_neck`, `out_channels`) and state_dict keys as the reference, so detection heads and checkpoints are interchangeable.

Everything between the input grid and the returned maps runs in the HIP library (channels-last, compute dtype): encoder kernels of
model.py, 1x1 lateral convs = `gemm_nt` with bias, top-down `nearest_upsample_add`, 3x3x3 output convs = `conv3d_k3_bias`, and
one transposing store to the NCDHW fp32 layout the heads consume.  Backward launches the matching dgrad / wgrad kernels.  No CPU
fallback: the ops raise without the library or with CPU tensors."""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
from torch import Tensor, nn

from . import ops
from .model import SWIN_CONFIGS, SwinTransformer_MAE3D_New, _EmbedFn, _gradbuf, _Packer


class _FPNFn(torch.autograd.Function):
    """lateral 1x1 -> top-down nearest add -> 3x3x3 convs (fpn.py:138-166) on channels-last tensors; outputs NCDHW fp32."""

    @staticmethod
    def forward(ctx, mod, *feats):
        fpn: "FPN" = mod
        pk, Co, n = fpn._pk, fpn.out_channels, len(feats)
        feats = [f.contiguous() for f in feats]
        lats = []
        for i, f in enumerate(feats):
            B, D, H, W, C = f.shape
            lat = ops.gemm_nt(f.view(-1, C), pk[f"l{i}.w"].view(Co, C), bias=fpn.lateral_convs[i].bias)
            lats.append(lat.view(B, D, H, W, Co))
        for i in range(n - 1, 0, -1):
            ops.nearest_upsample_add(lats[i], lats[i - 1])
        outs = []
        for i, lat in enumerate(lats):
            if ops.use_conv64(lat, Co) and f"f{i}.w64" in pk.views:   # LDS-halo kernel on 64-channel blocks (bias in its epilogue)
                y = ops.conv3d_k3_c64(lat, pk[f"f{i}.w64"], Co, bias=fpn.fpn_convs[i].bias)
            else:
                y = ops.conv3d_k3_bias(lat, pk[f"f{i}.w"], fpn.fpn_convs[i].bias, Co)
            outs.append(ops.ndhwc_to_ncdhw(y))
        ctx.fpn, ctx.saved = fpn, (feats, lats)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        fpn = ctx.fpn
        feats, lats = ctx.saved
        pk, Co, n = fpn._pk, fpn.out_channels, len(feats)
        dlats, keep = [], []   # keep: tensors the forked side stream still reads -- referenced until the join (the caching allocator knows nothing of that stream)
        for i, (lat, g) in enumerate(zip(lats, douts)):
            B, D, H, W, _ = lat.shape
            if g is None:
                dlats.append(torch.zeros_like(lat))
                continue
            dy = ops.ncdhw_to_ndhwc(g, lat.dtype)
            keep.append(dy)
            conv = fpn.fpn_convs[i]
            with ops.side_stream():
                ops.conv3d_k3_wgrad(dy, lat, _gradbuf(conv.weight))
                ops.bias_grad(dy.view(-1, Co), _gradbuf(conv.bias), B * D * H * W, Co)
            if ops.use_conv64(dy, Co) and f"f{i}.w64d" in pk.views:
                dlats.append(ops.conv3d_k3_c64(dy, pk[f"f{i}.w64d"], Co))
            else:
                dlats.append(ops.conv3d_k3(dy, pk[f"f{i}.wd"], Co))
        for i in range(1, n):   # d lat_i += adjoint of the nearest upsample of d lat_{i-1} (which is final by then)
            ops.nearest_upsample_add_bwd(dlats[i - 1], dlats[i])
        dfeats = []
        for i, (f, dl) in enumerate(zip(feats, dlats)):
            C = f.shape[-1]
            lconv = fpn.lateral_convs[i]
            keep.append(dl)
            dfeats.append(ops.gemm_nt(dl.view(-1, Co), pk[f"l{i}.wT"].view(C, Co)).view(f.shape))
            with ops.side_stream():
                ops.gemm_tn(dl.view(-1, Co), f.view(-1, C), _gradbuf(lconv.weight).view(Co, C), dbias=_gradbuf(lconv.bias))
        ops.join_side()
        del keep
        return (None, *dfeats)


class FPN(nn.Module):
    """Feature pyramid neck, 3-D (nerf_rpn/model/fpn.py).  Parameters live in `nn.Conv3d` holders so that names, shapes and the
    default initialisation equal the reference's; their forward is never used."""

    def __init__(self, in_channels: Sequence[int], out_channels: int, num_outs: int, start_level: int = 0, end_level: int = -1,
                 add_extra_convs=False, compute_dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        assert isinstance(in_channels, (list, tuple))
        if start_level != 0 or end_level not in (-1, len(in_channels)) or add_extra_convs or num_outs != len(in_channels):
            raise NotImplementedError("only the configuration of the Swin backbones (feature_extractor.py:1174) is built: "
                                      "start_level=0, end_level=-1, no extra levels")
        if out_channels % 8 or any(c % 8 for c in in_channels):
            raise NotImplementedError("channel counts must be multiples of 8")
        self.in_channels, self.out_channels = list(in_channels), out_channels
        self.num_ins, self.num_outs = len(in_channels), num_outs
        self.compute_dtype = compute_dtype
        self.lateral_convs = nn.ModuleList([nn.Conv3d(c, out_channels, 1) for c in in_channels])
        self.fpn_convs = nn.ModuleList([nn.Conv3d(out_channels, out_channels, 3, padding=1) for _ in in_channels])
        self._pk: Optional[_Packer] = None
        self._pk_key = None

    def init_weights(self):
        """fpn.py:128-133"""
        for m in self.modules():
            if isinstance(m, nn.Conv3d):
                nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)

    def _pack(self):
        ps = list(self.parameters())
        if not ps[0].is_cuda:
            raise RuntimeError("FPN (HIP) needs its parameters on a HIP device: call .cuda() first (no CPU fallback)")
        key = tuple(p.data_ptr() for p in ps)
        if self._pk is None or self._pk_key != key:
            P = _Packer()
            for i, (l, c) in enumerate(zip(self.lateral_convs, self.fpn_convs)):
                P.add(f"l{i}.w", l.weight, P.CAST)
                P.add(f"l{i}.wT", l.weight, P.TRANS)
                P.add(f"f{i}.w", c.weight, P.CONV_F)
                P.add(f"f{i}.wd", c.weight, P.CONV_D)
                if self.compute_dtype == torch.bfloat16 and c.weight.shape[0] % 64 == 0 and c.weight.shape[1] % 64 == 0:
                    P.add(f"f{i}.w64", c.weight, P.C64_F)
                    P.add(f"f{i}.w64d", c.weight, P.C64_D)
            P.build(self.compute_dtype, ps[0].device)
            P.split = None
            self._pk, self._pk_key = P, key
        self._pk.run()

    def forward_channels_last(self, feats: List[Tensor]):
        """feats: channels-last (B,D,H,W,C_i) tensors in the compute dtype (what the encoder kernels produce)"""
        assert len(feats) == len(self.in_channels)
        self._pack()
        return _FPNFn.apply(self, *feats)

    def forward(self, inputs):
        """reference signature: NCDHW tensors in, tuple of NCDHW fp32 tensors out (fpn.py:135-185)"""
        assert len(inputs) == len(self.in_channels)
        return self.forward_channels_last([_NCDHWToCL.apply(x, self.compute_dtype) for x in inputs])


class _NCDHWToCL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        return ops.ncdhw_to_ndhwc(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return ops.ndhwc_to_ncdhw(g.contiguous()), None


class SwinTransformer_FPN_Pretrained_Skip(nn.Module):
    """feature_extractor.py:1067-1187.  `checkpoint_path` is a file written by the MAE trainer ({"epoch","state_dict","train_args"},
    run_swin_mae3d.py:471-489; the reference's and this build's are interchangeable) and is loaded strictly before the decoders, the
    head and the mask token are deleted.  `backbone_type` / `compute_dtype` are additions (the reference hard-codes swin_s, fp32)."""

    def __init__(self, expand_dim: bool = True, out_channels: int = 256, resolution=160, checkpoint_path=None, is_eval=False,
                 backbone_type: str = "swin_s", compute_dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        self.out_channels = out_channels
        cfg = SWIN_CONFIGS[backbone_type]
        model = SwinTransformer_MAE3D_New(patch_size=[4, 4, 4], embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"],
                                          window_size=[4, 4, 4], stochastic_depth_prob=0.1, expand_dim=True, resolution=resolution,
                                          compute_dtype=compute_dtype)
        if not is_eval:
            assert checkpoint_path is not None and os.path.exists(checkpoint_path), "The checkpoint does not exist."
            checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
            model.load_state_dict(checkpoint["state_dict"])
        del model.decoder4
        del model.decoder3
        del model.decoder2
        del model.decoder1
        del model.out
        del model.mask_token
        dims = [cfg["embed_dim"] * 2 ** i if expand_dim else cfg["embed_dim"] for i in range(len(cfg["depths"]))]
        self.base = model
        self.fpn_neck = FPN(dims, out_channels, len(dims), compute_dtype=compute_dtype)

    def forward(self, x: Tensor):
        b = self.base
        b._ensure_ready(x.device)
        B, R = x.shape[0], x.shape[2]
        g = R // 4
        # patch_partition + pos_embed in one pass (feature_extractor.py:1179-1180), no masking
        tok = _EmbedFn.apply(b._anchor, b, x.float().contiguous(), None).view(B, g, g, g, b.embed_dim)
        feats = b.forward_encoder(tok)
        return self.fpn_neck.forward_channels_last(feats)
