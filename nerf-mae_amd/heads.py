"""Dense-prediction heads on the MAE-pretrained encoder + decoder: host-side mirrors of
`SwinTransformer_VoxelSR_Pretrained_Skip` (nerf_rpn/model/feature_extractor.py:1898-2244) and
`SwinTransformer_VoxelSemantics_Pretrained_Skip` (:2521-2848) with the reference's constructor arguments, attributes
(`base`, `encoder1`, `decoder1`, `voxel_out` / `sem_out`, `final_upsample`), state_dict keys, `forward(list of grids) -> pred (NCDHW fp32)`
and `loss_fn` / `forward_loss` contracts.  Every forward and backward op is a HIP kernel behind the C ABI: the base encoder and
decoder4..2 are the MAE path's blocks, `decoder1` is the same up-block kernel set with a skip connection, `encoder1` (a UnetResBlock on
the raw 4-channel grid) runs on an 8-channel zero-padded channels-last copy of the input, the 1x1x1 output conv runs at the decoder
resolution BEFORE the nearest upsampling (they commute) and the two losses are csrc/heads.hip.  No CPU fallback."""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import Tensor, nn

from . import ops
from .model import SWIN_CONFIGS, OutBlock3D, ResBlock3D, SwinTransformer_MAE3D_New, UpBlock3D, _EmbedFn, _gradbuf, _Packer


class UnetrBasicBlock3D(nn.Module):
    """UnetrBasicBlock(res_block=True) parameter holder (unetr_block.py:323-370): a UnetResBlock under `layer`"""

    def __init__(self, cin, cout):
        super().__init__()
        self.layer = ResBlock3D(cin, cout)


class _ResBlock8Fn(torch.autograd.Function):
    """encoder1: UnetResBlock(4 -> C) on the grid (unetr_block.py:57-71) from the 8-channel padded channels-last input; parameters only
    (the input is data: no input gradient)."""

    @staticmethod
    def forward(ctx, anchor, head, x8, B, S):
        blk = head.encoder1.layer
        pk = head._pk
        C = blk.conv2.weight.shape[0]
        V, dev = S ** 3, x8.device
        scratch = torch.empty((B, C, 2), dtype=torch.float64, device=dev)
        y1 = ops.conv3d_k3(x8, pk["e1.c1.w"], C).view(B * V, C)
        st1 = torch.empty((B, C, 2), device=dev)
        ops.instnorm_stats(y1, st1, scratch, B, V, C)
        a1 = torch.empty_like(y1)
        ops.instnorm_apply(y1, st1, a1, B, V, C)
        st2 = torch.empty((B, C, 2), device=dev)
        c48 = "e1.c2.wk" in pk.views
        if c48:
            y2 = ops.conv3d_k3_c48(a1.view(B, S, S, S, C), pk["e1.c2.wk"], stats_acc=scratch).view(B * V, C)
            ops.instnorm_finalize(scratch, st2, B, V, C)
        else:
            y2 = ops.conv3d_k3(a1.view(B, S, S, S, C), pk["e1.c2.w"], C).view(B * V, C)
            ops.instnorm_stats(y2, st2, scratch, B, V, C)
        y3 = ops.gemm_nt(x8.view(B * V, 8), pk["e1.c3.w"].view(C, 8))
        st3 = torch.empty((B, C, 2), device=dev)
        ops.instnorm_stats(y3, st3, scratch, B, V, C)
        out = torch.empty_like(y2)
        ops.instnorm_apply(y2, st2, out, B, V, C, r=y3, stats_r=st3, rmode=2)
        ctx.head, ctx.dims, ctx.c48 = head, (B, S, C), c48
        ctx.saved = (x8, y1, st1, a1, y2, st2, y3, st3, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        head = ctx.head
        blk, pk = head.encoder1.layer, head._pk
        B, S, C = ctx.dims
        V = S ** 3
        x8, y1, st1, a1, y2, st2, y3, st3, out = ctx.saved
        dev = x8.device
        dout = dout.contiguous()
        sums2 = torch.empty((B, C, 2), dtype=torch.float64, device=dev)
        sums3 = torch.empty_like(sums2)
        dy2, dy3 = torch.empty_like(y2), torch.empty_like(y3)
        ops.instnorm_bwd_reduce(dout, out, y2, st2, sums2, B, V, C, r=y3, stats_r=st3, sums_r=sums3, rmode=2)
        ops.instnorm_bwd_apply(dout, out, y2, st2, sums2, dy2, B, V, C, r=y3, stats_r=st3, sums_r=sums3, rmode=2, dr=dy3)
        if ctx.c48:
            da1 = ops.conv3d_k3_c48(dy2.view(B, S, S, S, C), pk["e1.c2.wkd"]).view(B * V, C)
            ops.conv3d_k3_c48_wgrad(dy2.view(B, S, S, S, C), a1.view(B, S, S, S, C), _gradbuf(blk.conv2.weight))
        else:
            da1 = ops.conv3d_k3(dy2.view(B, S, S, S, C), pk["e1.c2.wd"], C).view(B * V, C)
            ops.conv3d_k3_wgrad(dy2.view(B, S, S, S, C), a1.view(B, S, S, S, C), _gradbuf(blk.conv2.weight))
        sums1 = torch.empty_like(sums2)
        ops.instnorm_bwd_reduce(da1, None, y1, st1, sums1, B, V, C, rmode=0)
        dy1 = torch.empty_like(y1)
        ops.instnorm_bwd_apply(da1, None, y1, st1, sums1, dy1, B, V, C, rmode=0)
        # weight gradients against the 8-channel padded input, then the 4 real input channels are added into the parameters' gradients
        dW8 = torch.zeros((C, 8, 27), device=dev)
        ops.conv3d_k3_wgrad(dy1.view(B, S, S, S, C), x8, dW8)
        ops.add_cols_f32(dW8.view(C, 8 * 27), _gradbuf(blk.conv1.weight).view(C, 4 * 27), 4 * 27)
        dW3 = torch.zeros((C, 8), device=dev)
        ops.gemm_tn(dy3, x8.view(B * V, 8), dW3)
        ops.add_cols_f32(dW3, _gradbuf(blk.conv3.weight).view(C, 4), 4)
        for conv in (blk.conv1, blk.conv2, blk.conv3):   # a per-channel constant in front of an affine-free InstanceNorm cancels: exact zero gradient
            _gradbuf(conv.bias)
        return None, None, None, None, None


class _HeadFn(torch.autograd.Function):
    """1x1x1 output conv at the decoder resolution + nearest upsampling to the output resolution, NCDHW fp32 (UnetOutBlock after
    nn.Upsample in the reference, feature_extractor.py:2224-2229 / sem_out :2818; the two commute)."""

    @staticmethod
    def forward(ctx, d0, head, conv, B, R, Ro, inv_scale):
        pk = head._pk
        Co, C = conv.weight.shape[0], conv.weight.shape[1]
        Cop = (Co + 7) // 8 * 8
        bp = head._bias_pad
        bp[:Co].copy_(conv.bias.detach())                       # (a handful of floats: the GEMM epilogue reads Cop bias entries)
        y = ops.gemm_nt(d0, pk["out.w"].view(Cop, C), bias=bp)
        pred = ops.head_upsample_fwd(y, Co, B, R, Ro, inv_scale)
        ctx.head, ctx.conv, ctx.saved, ctx.dims = head, conv, (d0,), (B, R, Ro, inv_scale, Co, Cop, C)
        return pred

    @staticmethod
    def backward(ctx, dpred):
        head, conv = ctx.head, ctx.conv
        B, R, Ro, inv_scale, Co, Cop, C = ctx.dims
        (d0,) = ctx.saved
        g = ops.head_upsample_bwd(dpred.contiguous().float(), Cop, R, d0.dtype, inv_scale)
        dd0 = ops.gemm_nt(g, head._pk["out.wT"].view(C, Cop))
        ops.gemm_tn(g, d0, _gradbuf(conv.weight).view(Co, C), N=Co, dbias=_gradbuf(conv.bias))   # ragged N: only the Co real rows are written
        return dd0, None, None, None, None, None, None


class _SRLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        sums = torch.empty(2, dtype=torch.float64, device=pred.device)
        loss = torch.empty(1, device=pred.device)
        ops.voxel_sr_loss_fwd(pred, target, sums, loss)
        ctx.saved = (pred, target, sums)
        return loss[0]

    @staticmethod
    def backward(ctx, gl):
        pred, target, sums = ctx.saved
        dpred = torch.empty_like(pred)
        ops.voxel_sr_loss_bwd(pred, target, sums, float(gl), dpred)
        return dpred, None


class _CELossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, cw):
        B, K = logits.shape[:2]
        sums = torch.empty(2, dtype=torch.float64, device=logits.device)
        iou = torch.empty((B, K - 1, 3), dtype=torch.float64, device=logits.device)
        out = torch.empty(2, device=logits.device)
        ops.masked_ce_fwd(logits, labels, cw, sums, iou, out)
        ctx.saved = (logits, labels, cw, sums)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, gl, _gout):
        logits, labels, cw, sums = ctx.saved
        dl = torch.empty_like(logits)
        ops.masked_ce_bwd(logits, labels, cw, sums, float(gl), dl)
        return dl, None, None


class _VoxelHeadBase(nn.Module):
    backbone_type = "swin_s"    # hard-coded in the reference (feature_extractor.py:1936, 2546)

    def __init__(self, resolution, checkpoint_path, is_eval, patch_size, compute_dtype):
        super().__init__()
        cfg = SWIN_CONFIGS[self.backbone_type]
        model = SwinTransformer_MAE3D_New(patch_size=list(patch_size), embed_dim=cfg["embed_dim"], depths=cfg["depths"], num_heads=cfg["num_heads"],
                                          window_size=[4, 4, 4], stochastic_depth_prob=0.1, expand_dim=True, resolution=resolution, compute_dtype=compute_dtype)
        if not is_eval:   # feature_extractor.py:1988-2002: the MAE checkpoint is loaded into the FULL model before the decoder head is removed
            import os
            assert os.path.exists(checkpoint_path), "The checkpoint does not exist."
            checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
            model.load_state_dict(checkpoint["state_dict"])
        del model.decoder1
        del model.out
        del model.mask_token
        E = model.embed_dim
        self.patch_size, self.input_resolution, self.compute_dtype = list(patch_size), resolution, compute_dtype
        self.encoder1 = UnetrBasicBlock3D(4, E // 2)
        self.decoder1 = UpBlock3D(E, E // 2, 4, use_skip=True)
        self.base = model
        self._pk = None
        self._pk_key = None

    def _pack(self):
        ps = [p for n, p in self.named_parameters() if not n.startswith("base.")]
        if not ps[0].is_cuda:
            raise RuntimeError("the voxel heads (HIP) need their parameters on a HIP device: call .cuda() first (no CPU fallback)")
        key = tuple(p.data_ptr() for p in ps)
        if self._pk is None or self._pk_key != key:
            P = _Packer()
            e1, d, oc = self.encoder1.layer, self.decoder1, self._out_conv()
            P.add("e1.c1.w", e1.conv1.weight, P.PAD_CIN8)
            P.add("e1.c2.w", e1.conv2.weight, P.CONV_F)
            P.add("e1.c2.wd", e1.conv2.weight, P.CONV_D)
            if self.compute_dtype == torch.bfloat16 and tuple(e1.conv2.weight.shape[:2]) == (48, 48):
                P.add("e1.c2.wk", e1.conv2.weight, P.C48_F)
                P.add("e1.c2.wkd", e1.conv2.weight, P.C48_D)
            P.add("e1.c3.w", e1.conv3.weight, P.PAD_CIN8)
            k = "decoder1."
            d._key = k
            P.add(k + "t.w", d.transp_conv.weight, P.CONVT_F)
            P.add(k + "t.wd", d.transp_conv.weight, P.CONVT_D)
            for cn in ("c1", "c2"):
                conv = getattr(d.conv_block, "conv" + cn[1])
                P.add(k + cn + ".w", conv.weight, P.CONV_F)
                P.add(k + cn + ".wd", conv.weight, P.CONV_D)
            P.add(k + "c3.w", d.conv_block.conv3.weight, P.CAST)
            P.add(k + "c3.wT", d.conv_block.conv3.weight, P.TRANS)
            P.add("out.w", oc.weight, P.PAD_ROWS)
            P.add("out.wT", oc.weight, P.PAD_ROWS_T)
            P.build(self.compute_dtype, ps[0].device)
            P.split = None
            d._pk = P
            self._pk, self._pk_key = P, key
            self._bias_pad = torch.zeros((oc.weight.shape[0] + 7) // 8 * 8, device=ps[0].device)
        self._pk.run()

    def _dec0(self, x: List[Tensor]):
        """everything up to the last decoder level (feature_extractor.py:2199-2222 / 2797-2817), channels-last [B*R^3][E/2]"""
        b = self.base
        device = b.pos_embed.device
        b._ensure_ready(device)
        self._pack()
        xb, _ = b.transform(x, device)
        B, R = xb.shape[0], self.input_resolution
        g = R // 4
        x8 = ops.grid_to_cl8(xb, self.compute_dtype)
        enc1 = _ResBlock8Fn.apply(b._anchor, self, x8, B, R)
        tok = _EmbedFn.apply(b._anchor, b, xb, None).view(B, g, g, g, b.embed_dim)      # + pos_embed inside the LayerNorm kernel, no mask
        feats = b.forward_encoder(tok)
        b._packer.join()
        d = b.decoder4(feats[3], feats[2])
        d = b.decoder3(d, feats[1])
        d = b.decoder2(d, feats[0])
        dec0 = self.decoder1(d, enc1.view(B, R, R, R, -1))
        return dec0.reshape(B * R ** 3, -1), B

    def transform(self, x, resolution=160):
        """pad_tensor semantics (torch_utils.py:56-90): list of (C,W,L,H) -> (B,C,res,res,res) fp32 on the device, zero padded"""
        dev = self.base.pos_embed.device
        out = torch.zeros((len(x), x[0].shape[0], resolution, resolution, resolution), dtype=torch.float32, device=dev)
        for i, t in enumerate(x):
            a0, a1, a2 = t.shape[1:]
            out[i, :, :a0, :a1, :a2] = t.to(device=dev, dtype=torch.float32, non_blocking=True)
        return out


class SwinTransformer_VoxelSR_Pretrained_Skip(_VoxelHeadBase):
    """feature_extractor.py:1898-2244.  forward(list of (4,W,L,H) grids) -> pred (B,4,out_res^3); loss_fn(list of grids at the output
    resolution, pred) -> masked RGB MSE."""

    def __init__(self, expand_dim: bool = True, out_channels: int = 256, resolution=160, out_resolution=256, decoder_embed_dim: int = 768,
                 checkpoint_path=None, is_eval=False, patch_size=[4, 4, 4], compute_dtype: torch.dtype = torch.bfloat16):
        super().__init__(resolution, checkpoint_path, is_eval, patch_size, compute_dtype)
        self.out_channels, self.output_resolution = out_channels, out_resolution
        scale = {256: 1.6, 384: 2.4}[out_resolution]          # feature_extractor.py:2014-2017
        self.final_upsample = nn.Upsample(scale_factor=scale)   # attribute contract; the kernel applies the same index rule
        self.voxel_out = OutBlock3D(self.base.embed_dim // 2, 4)

    def _out_conv(self):
        return self.voxel_out.conv

    def forward(self, x: List[Tensor]) -> Tensor:
        d0, B = self._dec0(x)
        R = self.input_resolution
        scale = float(self.final_upsample.scale_factor)
        Ro = int(R * scale)       # floor(in * scale_factor), nn.Upsample's output size (256 / 384 for a 160^3 input)
        inv = float(torch.tensor(1.0 / scale, dtype=torch.float32))
        return _HeadFn.apply(d0, self, self.voxel_out.conv, B, R, Ro, inv)

    def forward_loss(self, x, pred, is_eval=False):
        return _SRLossFn.apply(pred, self.transform(x, resolution=pred.shape[2]))

    def loss_fn(self, x, pred):
        return self.forward_loss(x, pred)

    def output_metrics(self, x, pred):
        """MSE / PSNR over the RGB entries of occupied voxels (feature_extractor.py:2162-2183; evaluation bookkeeping on the outputs)"""
        with torch.no_grad():
            mse = float(self.forward_loss(x, pred)) / 3.0      # loss_rgb divides by the voxel count, mse by the entry count
        import math
        return {"MSE": mse, "PSNR": -10.0 * math.log10(mse)}


class SwinTransformer_VoxelSemantics_Pretrained_Skip(_VoxelHeadBase):
    """feature_extractor.py:2521-2848.  forward(list of (4,W,L,H) grids) -> class logits (B,K,res^3); loss_fn(list of (1,W,L,H) label grids,
    pred) -> (loss, sem_ce_loss, mean soft IoU)."""

    def __init__(self, expand_dim: bool = True, out_channels: int = 19, resolution=160, decoder_embed_dim: int = 768, checkpoint_path=None,
                 is_eval=False, patch_size=[4, 4, 4], class_weights: Optional[Tensor] = None, compute_dtype: torch.dtype = torch.bfloat16):
        super().__init__(resolution, checkpoint_path, is_eval, patch_size, compute_dtype)
        if out_channels > 32:
            raise ValueError("the cross-entropy kernel holds the class logits of a voxel in registers: out_channels <= 32")
        self.out_channels = out_channels
        self.class_weights = class_weights
        self.sem_out = OutBlock3D(self.base.embed_dim // 2, out_channels)

    def _out_conv(self):
        return self.sem_out.conv

    def forward(self, x: List[Tensor]) -> Tensor:
        d0, B = self._dec0(x)
        R = self.input_resolution
        return _HeadFn.apply(d0, self, self.sem_out.conv, B, R, R, 1.0)

    def forward_loss(self, x, pred, is_eval=False):
        labels = self.transform(x, resolution=self.input_resolution)[:, 0].contiguous()
        cw = None if self.class_weights is None else self.class_weights.to(device=pred.device, dtype=torch.float32)
        loss, out = _CELossFn.apply(pred, labels, cw)
        return loss, loss, out[1]

    def loss_fn(self, x, pred):
        return self.forward_loss(x, pred)
