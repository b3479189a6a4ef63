"""CPU oracle for the NeRF-MAE 3-D Swin MAE hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch fp32 (CPU) restatement of the reference algorithm
`SwinTransformer_MAE3D_New` (reference: nerf_mae/model/mae/swin_mae3d.py:1067-1599,
unetr_block.py:23-200, torch_utils.py:5-90).  It exists to *check* the HIP product
path; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import it.  The product package (`nerf-mae_amd/`) never imports anything from
`oracle/` and has no CPU fallback.

Parity pin: `oracle/gen_golden.py` imports the real reference in the build container
and writes `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks this restatement
against those vectors (<=1e-5 abs/rel in fp32).  The torchvision 0.13.1 pieces the
reference uses (`StochasticDepth("row")`, `MLP`, `Permute`, swin_mae3d.py:5-6) are not
in /root/reference; their published semantics are restated below ("parity unpinned"
at that third-party boundary -- see DESIGN.md).

State-dict keys and shapes are identical to the reference so checkpoints interchange.
"""
from __future__ import annotations

import math
import random
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor, nn

WS = 4  # window edge (reference fixes window_size=[4,4,4], run_swin_mae3d.py:400-411)

SWIN_CONFIGS = {
    # run_swin_mae3d.py:378-399 ; swin_b uses the SURVEY 8(c) defined deviation (heads 4,8,16,32)
    "swin_t": dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24]),
    "swin_s": dict(embed_dim=96, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24]),
    "swin_b": dict(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32]),
    "swin_l": dict(embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48]),
}


# ----------------------------------------------------------------------------------------------
# fixed tables
# ----------------------------------------------------------------------------------------------
def sincos_1d(dim: int, pos: np.ndarray) -> np.ndarray:
    """torch_utils.py:34-53 -- [sin(p*w) | cos(p*w)], w_d = 10000^(-d/(dim/2)), float64."""
    omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
    ang = pos.reshape(-1).astype(np.float64)[:, None] * omega[None, :]
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)


def sincos_pos_embed_3d(embed_dim: int, g: int, pad_to: Optional[int] = None) -> np.ndarray:
    """torch_utils.py:5-31.  Token (i,j,k): thirds encode (j, i, k) -- the reference's
    np.meshgrid(w,h,l) default 'xy' indexing swaps the first two axes."""
    third = embed_dim // 3
    i, j, k = np.meshgrid(np.arange(g), np.arange(g), np.arange(g), indexing="ij")
    emb = np.concatenate([sincos_1d(third, j), sincos_1d(third, i), sincos_1d(third, k)], axis=1)
    if pad_to is not None and emb.shape[1] < pad_to:  # swin_b defined deviation: zero-pad 126 -> 128
        emb = np.concatenate([emb, np.zeros((emb.shape[0], pad_to - emb.shape[1]))], axis=1)
    return emb.reshape(1, g, g, g, -1)


def rel_pos_index(ws: int = WS) -> Tensor:
    """swin_mae3d.py:257-280 -- index = (dh+ws-1)*(2ws-1)^2 + (dw+ws-1)*(2ws-1) + (dd+ws-1), [ws^6] int64."""
    c = torch.stack(torch.meshgrid(*[torch.arange(ws)] * 3, indexing="ij")).flatten(1)  # 3, ws^3
    rel = (c[:, :, None] - c[:, None, :]) + (ws - 1)
    m = 2 * ws - 1
    return (rel[0] * m * m + rel[1] * m + rel[2]).flatten()


def shift_region_ids(P: Sequence[int], shift: Sequence[int], ws: int = WS) -> Tensor:
    """swin_mae3d.py:126-147.  Region label per padded coordinate, replaying the reference's
    slice assignments (including the degenerate shift==0 axis where `[-ws:-0]` is empty and
    `[-0:None]` covers the whole axis)."""
    ids = torch.zeros(tuple(P))
    cnt = 0
    sl = [((0, -ws), (-ws, -s), (-s, None)) for s in shift]
    for a in sl[0]:
        for b in sl[1]:
            for c in sl[2]:
                ids[a[0]:a[1], b[0]:b[1], c[0]:c[1]] = cnt
                cnt += 1
    return ids


# ----------------------------------------------------------------------------------------------
# functional ops (also used one-by-one by the per-kernel parity tests)
# ----------------------------------------------------------------------------------------------
def window_partition(x: Tensor, ws: int = WS) -> Tensor:
    B, H, W, D, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, D // ws, ws, C)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(-1, ws ** 3, C)


def window_reverse(xw: Tensor, B: int, H: int, W: int, D: int, ws: int = WS) -> Tensor:
    C = xw.shape[-1]
    x = xw.view(B, H // ws, W // ws, D // ws, ws, ws, ws, C)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, H, W, D, C)


def window_attention(x: Tensor, qkv_w: Tensor, qkv_b: Tensor, proj_w: Tensor, proj_b: Tensor,
                     bias_table: Tensor, heads: int, shift: Sequence[int], ws: int = WS) -> Tensor:
    """swin_mae3d.py:27-197 (non-cosine branch).  x: (B,H,W,D,C) already layer-normed."""
    B, H, W, D, C = x.shape
    pad = [(ws - s % ws) % ws for s in (H, W, D)]
    x = F.pad(x, (0, 0, 0, pad[2], 0, pad[1], 0, pad[0]))
    P = x.shape[1:4]
    shift = [0 if ws >= P[a] else int(shift[a]) for a in range(3)]
    if sum(shift) > 0:
        x = torch.roll(x, shifts=[-s for s in shift], dims=(1, 2, 3))
    xw = window_partition(x, ws)  # (B*nW, 64, C)
    nW = xw.shape[0] // B
    N = ws ** 3
    hd = C // heads
    qkv = F.linear(xw, qkv_w, qkv_b).view(-1, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = bias_table[rel_pos_index(ws)].view(N, N, heads).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if sum(shift) > 0:
        ids = window_partition(shift_region_ids(P, shift, ws)[None, ..., None], ws).view(nW, N)
        m = (ids[:, None, :] - ids[:, :, None]) != 0
        am = torch.zeros(nW, N, N).masked_fill(m, -100.0)
        attn = (attn.view(B, nW, heads, N, N) + am[None, :, None]).view(-1, heads, N, N)
    attn = attn.softmax(-1)
    o = (attn @ v).transpose(1, 2).reshape(-1, N, C)
    o = F.linear(o, proj_w, proj_b)
    o = window_reverse(o, B, P[0], P[1], P[2], ws)
    if sum(shift) > 0:
        o = torch.roll(o, shifts=list(shift), dims=(1, 2, 3))
    return o[:, :H, :W, :D].contiguous()


def patch_merge_gather(x: Tensor) -> Tensor:
    """swin_mae3d.py:390-402 -- concat order index = h_off + 2*w_off + 4*d_off."""
    H, W, D = x.shape[-4:-1]
    x = F.pad(x, (0, 0, 0, D % 2, 0, W % 2, 0, H % 2))
    parts = [x[..., a::2, b::2, c::2, :] for c in (0, 1) for b in (0, 1) for a in (0, 1)]
    return torch.cat(parts, -1)


def draw_block_mask(g: Sequence[int], p_remove: float, block: int = 4, rng=random) -> Tensor:
    """swin_mae3d.py:1366-1373 -- one Bernoulli draw per 4x4x4-token block, raster (h,w,d),
    python `random`.  Returns float mask (gh,gw,gd) with 1 = removed."""
    m = torch.zeros(tuple(g))
    for h in range(0, g[0] - block + 1, block):
        for w in range(0, g[1] - block + 1, block):
            for d in range(0, g[2] - block + 1, block):
                if rng.random() < p_remove:
                    m[h:h + block, w:w + block, d:d + block] = 1
    return m


def pad_grid(t: Tensor, R: int) -> Tuple[Tensor, Tensor]:
    """torch_utils.py:56-90 -- zero-pad (4,W,L,H) at the high end to (4,R,R,R); ones-mask of valid voxels."""
    padw = (0, R - t.shape[3], 0, R - t.shape[2], 0, R - t.shape[1])
    return F.pad(t, padw)[None], F.pad(torch.ones_like(t), padw)[None]


def patchify(x: Tensor, p: int = 4) -> Tensor:
    """swin_mae3d.py:1384-1394 -- (N,4,R,R,R) -> (N,g,g,g,p^3,4)."""
    N, C, R = x.shape[0], x.shape[1], x.shape[2]
    g = R // p
    x = x.reshape(N, C, g, p, g, p, g, p).permute(0, 2, 4, 6, 3, 5, 7, 1)
    return x.reshape(N, g, g, g, p ** 3, C)


def mae_loss(x: Tensor, pred: Tensor, valid: Tensor, token_mask: Tensor, p: int = 4):
    """swin_mae3d.py:1513-1549.  x,pred,valid: (N,4,R,R,R); token_mask: (N,g,g,g,1) float {0,1}."""
    tgt, prd = patchify(x, p), patchify(pred, p)
    vm = patchify(valid, p)[..., 0].int()  # (N,g,g,g,64)
    m_rm = (vm * token_mask).unsqueeze(-1).int()
    t_rgb, t_a = tgt[..., :3], tgt[..., 3:]
    p_rgb, p_a = prd[..., :3], prd[..., 3:]
    occ = t_a > 0.01
    l_rgb = (((p_rgb - t_rgb) ** 2) * occ).sum() / occ.sum()
    l_a = (((torch.sigmoid(p_a) - t_a) ** 2) * m_rm).sum() / m_rm.sum()
    return l_rgb + l_a, l_rgb, l_a, prd, occ, tgt


# ----------------------------------------------------------------------------------------------
# modules (parameter names == reference)
# ----------------------------------------------------------------------------------------------
class _Permute(nn.Module):
    def __init__(self, dims):
        super().__init__()
        self.dims = dims

    def forward(self, x):
        return x.permute(self.dims)


class RowStochasticDepth(nn.Module):
    """torchvision 0.13.1 StochasticDepth(p, "row"): train-only, noise shape [B,1,..],
    bernoulli_(1-p) then div_(1-p)."""

    def __init__(self, p: float):
        super().__init__()
        self.p = p

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        keep = 1.0 - self.p
        noise = torch.empty([x.shape[0]] + [1] * (x.ndim - 1), dtype=x.dtype, device=x.device).bernoulli_(keep)
        if keep > 0:
            noise.div_(keep)
        return x * noise


class WindowAttention3D(nn.Module):
    def __init__(self, dim, heads, shift, ws=WS):
        super().__init__()
        self.heads, self.shift, self.ws = heads, list(shift), ws
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 3, heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.register_buffer("relative_position_index", rel_pos_index(ws))

    def forward(self, x):
        return window_attention(x, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias,
                                self.relative_position_bias_table, self.heads, self.shift, self.ws)


class SwinBlock3D(nn.Module):
    """swin_mae3d.py:310-369."""

    def __init__(self, dim, heads, shift, sd_prob, mlp_ratio=4.0, eps=1e-5):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = WindowAttention3D(dim, heads, shift)
        self.stochastic_depth = RowStochasticDepth(sd_prob)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        hid = int(dim * mlp_ratio)
        # torchvision MLP == Sequential(Linear, GELU, Dropout, Linear, Dropout): param slots 0 and 3
        self.mlp = nn.Sequential(nn.Linear(dim, hid), nn.GELU(), nn.Dropout(0.0), nn.Linear(hid, dim), nn.Dropout(0.0))

    def forward(self, x):
        x = x + self.stochastic_depth(self.attn(self.norm1(x)))
        return x + self.stochastic_depth(self.mlp(self.norm2(x)))


class PatchMerging3D(nn.Module):
    """swin_mae3d.py:372-414."""

    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.reduction = nn.Linear(8 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(8 * dim, eps=eps)

    def forward(self, x):
        return self.reduction(self.norm(patch_merge_gather(x)))


class ResBlock3D(nn.Module):
    """unetr_block.py:23-71."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = nn.Conv3d(cin, cout, 3, 1, 1)
        self.conv2 = nn.Conv3d(cout, cout, 3, 1, 1)
        self.has_proj = cin != cout
        if self.has_proj:
            self.conv3 = nn.Conv3d(cin, cout, 1, 1)

    def forward(self, x):
        inorm = lambda t: F.instance_norm(t, eps=1e-5)
        o = F.leaky_relu(inorm(self.conv1(x)), 0.01)
        o = inorm(self.conv2(o))
        r = inorm(self.conv3(x)) if self.has_proj else x
        return F.leaky_relu(o + r, 0.01)


class UpBlock3D(nn.Module):
    """unetr_block.py:119-200."""

    def __init__(self, cin, cout, k, use_skip=True):
        super().__init__()
        self.use_skip = use_skip
        self.transp_conv = nn.ConvTranspose3d(cin, cout, k, stride=k)
        self.conv_block = ResBlock3D(2 * cout if use_skip else cout, cout)

    def forward(self, x, skip=None):
        o = self.transp_conv(x)
        if self.use_skip:
            o = torch.cat((o, skip), 1)
        return self.conv_block(o)


class OutBlock3D(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, 1)

    def forward(self, x):
        return self.conv(x)


class MAE3DOracle(nn.Module):
    """CPU fp32 restatement of SwinTransformer_MAE3D_New (swin_mae3d.py:1067-1599)."""

    def __init__(self, patch_size=(4, 4, 4), embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24),
                 window_size=(4, 4, 4), mlp_ratio=4.0, stochastic_depth_prob=0.1, out_channels=4,
                 input_ch_dim=4, masking_prob=0.5, resolution=160, pad_pos_embed=False, **_unused):
        super().__init__()
        assert tuple(window_size) == (WS,) * 3 and tuple(patch_size) == (4, 4, 4)
        self.patch_size, self.embed_dim = list(patch_size), embed_dim
        self.masking_prob, self.resolution, self.out_channels = masking_prob, resolution, out_channels
        self.patch_partition = nn.Sequential(
            nn.Conv3d(input_ch_dim, embed_dim, 4, 4), _Permute([0, 2, 3, 4, 1]), nn.LayerNorm(embed_dim, eps=1e-5))
        self.stages = nn.ModuleList()
        total, bid = sum(depths), 0
        for s, depth in enumerate(depths):
            dim = embed_dim * 2 ** s
            mods: List[nn.Module] = [PatchMerging3D(dim // 2)] if s > 0 else []
            for i in range(depth):
                sd = stochastic_depth_prob * float(bid) / (total - 1)
                mods.append(SwinBlock3D(dim, num_heads[s], [0] * 3 if i % 2 == 0 else [WS // 2] * 3, sd, mlp_ratio))
                bid += 1
            self.stages.append(nn.Sequential(*mods))
        E = embed_dim
        self.decoder4 = UpBlock3D(8 * E, 4 * E, 2)
        self.decoder3 = UpBlock3D(4 * E, 2 * E, 2)
        self.decoder2 = UpBlock3D(2 * E, E, 2)
        self.decoder1 = UpBlock3D(E, E // 2, 4, use_skip=False)
        self.out = OutBlock3D(E // 2, out_channels)
        g = resolution // patch_size[0]
        self.num_patches = g
        self.pos_embed = nn.Parameter(torch.zeros(1, g, g, g, E), requires_grad=False)
        self.mask_token = nn.Parameter(torch.zeros(E))
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        pe = sincos_pos_embed_3d(E, g, pad_to=E if pad_pos_embed else None)
        self.pos_embed.data.copy_(torch.from_numpy(pe).float())
        nn.init.normal_(self.mask_token, std=0.02)

    # -- pieces -------------------------------------------------------------------------------
    def embed(self, x: Tensor, block_mask: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
        """swin_mae3d.py:1455-1463: patch embed + pos-embed + mask-token replacement."""
        t = self.patch_partition(x) + self.pos_embed.detach()
        B, g0, g1, g2, _ = t.shape
        if block_mask is None:
            block_mask = draw_block_mask((g0, g1, g2), self.masking_prob)
        m = block_mask.to(t.dtype)[None, ..., None].expand(B, -1, -1, -1, 1)
        t = torch.where(m.bool(), self.mask_token.to(t.dtype).view(1, 1, 1, 1, -1), t)
        return t, m.contiguous()

    def encode(self, t: Tensor) -> List[Tensor]:
        feats = []
        for st in self.stages:
            t = st(t)
            feats.append(t)
        return feats  # channels-last (B,s,s,s,C)

    def decode(self, feats: List[Tensor]) -> Tensor:
        f = [z.permute(0, 4, 1, 2, 3).contiguous() for z in feats]
        d = self.decoder4(f[3], f[2])
        d = self.decoder3(d, f[1])
        d = self.decoder2(d, f[0])
        return self.out(self.decoder1(d))

    def forward(self, x: List[Tensor], is_eval: bool = False, block_mask: Optional[Tensor] = None,
                return_pred: bool = False):
        grids, valids = zip(*[pad_grid(t, self.resolution) for t in x])
        xb, vb = torch.cat(grids, 0), torch.cat(valids, 0)
        tok, tmask = self.embed(xb, block_mask)
        pred = self.decode(self.encode(tok))
        loss, l_rgb, l_a, prd, occ, tgt = mae_loss(xb, pred, vb, tmask, self.patch_size[0])
        if return_pred:
            return loss, l_rgb, l_a, pred
        if is_eval:
            return loss, l_rgb, l_a, prd, occ, tgt
        return loss, l_rgb, l_a

    # nerf_rpn contract (feature_extractor.py:1176-1187): encoder-only NCDHW feature list
    def encoder_features(self, xb: Tensor) -> List[Tensor]:
        t = self.patch_partition(xb) + self.pos_embed.detach()
        return [z.permute(0, 4, 1, 2, 3).contiguous() for z in self.encode(t)]


class FPNOracle(nn.Module):
    """CPU restatement of the FPN neck as SwinTransformer_FPN_Pretrained_Skip configures it (nerf_rpn/model/fpn.py:57-166 with
    start_level=0, end_level=-1, add_extra_convs=False, num_outs=len(in_channels), upsample nearest): 1x1 lateral convs, top-down
    `laterals[i-1] += F.interpolate(laterals[i], size=prev_shape, mode="nearest")` (fpn.py:150-159), 3x3x3 output convs."""

    def __init__(self, in_channels: Sequence[int], out_channels: int, num_outs: int):
        super().__init__()
        assert num_outs == len(in_channels)
        self.in_channels, self.out_channels = list(in_channels), out_channels
        self.lateral_convs = nn.ModuleList([nn.Conv3d(c, out_channels, 1) for c in in_channels])
        self.fpn_convs = nn.ModuleList([nn.Conv3d(out_channels, out_channels, 3, padding=1) for _ in in_channels])

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        lat = [l(x) for l, x in zip(self.lateral_convs, inputs)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode="nearest")
        return tuple(c(x) for c, x in zip(self.fpn_convs, lat))


class FPNSkipOracle(nn.Module):
    """CPU restatement of SwinTransformer_FPN_Pretrained_Skip (nerf_rpn/model/feature_extractor.py:1067-1187): the MAE-pretrained
    encoder (decoders, head and mask token deleted, :1159-1164) + FPN neck; forward(x (B,4,R,R,R)) -> 4 NCDHW maps (:1176-1187)."""

    def __init__(self, out_channels: int = 256, resolution: int = 160, backbone_type: str = "swin_s", **kw):
        super().__init__()
        base = build_oracle(backbone_type, resolution=resolution, stochastic_depth_prob=kw.pop("stochastic_depth_prob", 0.1), **kw)
        del base.decoder4, base.decoder3, base.decoder2, base.decoder1, base.out, base.mask_token
        self.base = base
        E = base.embed_dim
        self.out_channels = out_channels
        self.fpn_neck = FPNOracle([E, 2 * E, 4 * E, 8 * E], out_channels, 4)

    def forward(self, x: Tensor):
        return self.fpn_neck(self.base.encoder_features(x))


def build_oracle(name: str = "swin_t", **kw) -> MAE3DOracle:
    cfg = dict(SWIN_CONFIGS[name])
    if name == "swin_b":
        kw.setdefault("pad_pos_embed", True)
    cfg.update(kw)
    return MAE3DOracle(patch_size=[4] * 3, window_size=[4] * 3, **cfg)


# ----------------------------------------------------------------------------------------------
# deterministic formula-filled tensors (no weight blobs are committed)
# ----------------------------------------------------------------------------------------------
def _name_phase(name: str) -> float:
    h = 0
    for ch in name:
        h = (h * 131 + ord(ch)) % 1000003
    return (h % 6283) / 1000.0


def formula_tensor(name: str, shape, scale: float = 1.0, offset: float = 0.0) -> Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(n, dtype=np.float64)
    ph = _name_phase(name)
    v = np.sin(0.37 * i + ph) * 0.6 + np.sin(0.011 * i * (1.0 + ph / 7.0) + 2.0 * ph) * 0.4
    return torch.from_numpy((offset + scale * v).astype(np.float32)).reshape(tuple(shape))


def formula_fill_(module: nn.Module) -> None:
    """Overwrite every trainable parameter with a name-seeded closed-form pattern whose scale
    mimics the real init (so activations stay O(1))."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if not p.requires_grad:
                continue
            if name.endswith("norm.weight") or ".norm1.weight" in name or ".norm2.weight" in name or name == "patch_partition.2.weight":
                p.copy_(formula_tensor(name, p.shape, 0.2, 1.0))
            elif name.endswith(".bias") or name == "mask_token":
                p.copy_(formula_tensor(name, p.shape, 0.05))
            elif "relative_position_bias_table" in name:
                p.copy_(formula_tensor(name, p.shape, 0.5))
            else:
                fan_in = int(np.prod(p.shape[1:])) if p.ndim > 1 else p.numel()
                if "transp_conv.weight" in name:
                    fan_in = p.shape[0]
                p.copy_(formula_tensor(name, p.shape, 1.2 / math.sqrt(max(fan_in, 1))))


def seeded_reference_init_(module: nn.Module, seed: int = 0) -> None:
    """The reference's initialisation DISTRIBUTIONS (swin_mae3d.py:1272-1276,1312: trunc_normal(0.02) Linear weights / zero biases /
    normal(0.02) mask token; PyTorch defaults elsewhere: kaiming-uniform(a=sqrt 5) conv weights and U(-1/sqrt(fan_in), ..) conv biases,
    LayerNorm ones/zeros, trunc_normal(0.02) relative-position tables), drawn from one generator PER PARAMETER seeded by (seed, name):
    the reference model, the oracle and the HIP model get bit-identical weights whatever their construction order."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if not p.requires_grad:
                continue
            g = torch.Generator().manual_seed(seed * 1000003 + int(_name_phase(name) * 1000) + len(name))
            is_norm = name.endswith("norm.weight") or ".norm1." in name or ".norm2." in name or name.startswith("patch_partition.2.") or name.endswith(".norm.bias")
            if is_norm:
                p.fill_(1.0 if name.endswith("weight") else 0.0)
            elif name == "mask_token":
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            elif "relative_position_bias_table" in name:
                p.copy_(torch.nn.init.trunc_normal_(torch.empty(p.shape), std=0.02, generator=g))
            elif "conv" in name or name.startswith("patch_partition.0."):   # Conv3d / ConvTranspose3d: PyTorch defaults
                w_shape = p.shape if p.ndim > 1 else None
                if w_shape is not None:
                    fan_in = int(np.prod(p.shape[1:]))          # (ConvTranspose3d: weight is (Cin, Cout, k,k,k); PyTorch uses size(1)*k^3 as well)
                    b = 1.0 / math.sqrt(fan_in)
                    p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * b)
                else:   # bias: bound from the layer's weight fan-in
                    wname = name[:-4] + "weight"
                    w = dict(module.named_parameters())[wname]
                    b = 1.0 / math.sqrt(int(np.prod(w.shape[1:])))
                    p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * b)
            elif p.ndim == 2:            # nn.Linear weights
                p.copy_(torch.nn.init.trunc_normal_(torch.empty(p.shape), std=0.02, generator=g))
            else:                        # Linear biases
                p.zero_()


def synthetic_grid(shape=(160, 160, 160), seed: int = 0) -> Tensor:
    """SURVEY 8(d) synthetic RGB-sigma grid: RGB ~ U[0,1); alpha = clip(1-exp(-exp(sigma)/100),0,1),
    sigma ~ N(0,3^2) inside a 60% sub-box, -10 outside (mirrors nerf_rpn/datasets.py:247-248)."""
    g = torch.Generator().manual_seed(seed)
    rgb = torch.rand((3,) + tuple(shape), generator=g)
    sigma = torch.full(tuple(shape), -10.0)
    lo = [int(0.2 * s) for s in shape]
    hi = [int(0.8 * s) for s in shape]
    box = torch.randn([h - l for l, h in zip(lo, hi)], generator=g) * 3.0
    sigma[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = box
    alpha = torch.clamp(1.0 - torch.exp(-torch.exp(sigma) / 100.0), 0.0, 1.0)
    return torch.cat([rgb, alpha[None]], 0)


# ---------------------------------------------------------------------------------------------------------------------
# input pipeline restatement; pinned by tests/golden/g11_input_pipeline.npz (oracle/gen_golden_input.py)
# ---------------------------------------------------------------------------------------------------------------------
GRID_ROT, GRID_FLIP0, GRID_FLIP1, GRID_DENSITY = 1, 2, 4, 8


def density_to_alpha(density):
    """nerf_rpn/datasets.py:247-248"""
    import numpy as np
    return np.clip(1.0 - np.exp(-np.exp(density) / 100.0), 0.0, 1.0)


def draw_augmentation(flip_prob: float, rotate_prob: float, rng=random) -> int:
    """order of the random draws in augment_rpn_inputs with boxes=None, z_up=True (datasets.py:198-233)"""
    flags = 0
    if rng.random() < rotate_prob:
        flags |= GRID_ROT
    for bit in (GRID_FLIP0, GRID_FLIP1):
        if rng.random() < flip_prob:
            flags |= bit
    return flags


def prepare_grid(scene, R: int, flags: int = 0):
    """stored scene (W,L,H,4) float32|uint8 numpy -> ((4,R,R,R) float32 tensor, extents): datasets.py:88-101 (density->alpha when
    GRID_DENSITY is set and the scene is float, uint8 -> /255, channels first), :198-233 (rotation = transpose(1,2) + flip(1), flips),
    torch_utils.py:56-90 (zero padding at the high end)."""
    import numpy as np
    g = np.array(scene, copy=True)
    if g.dtype != np.uint8 and flags & GRID_DENSITY:
        g[..., -1] = density_to_alpha(g[..., -1])
    t = torch.from_numpy(np.ascontiguousarray(np.transpose(g, (3, 0, 1, 2))))
    if t.dtype == torch.uint8:
        t = t.float() / 255.0
    if flags & GRID_ROT:
        t = torch.flip(torch.transpose(t, 1, 2), [1])
    if flags & GRID_FLIP0:
        t = t.flip(dims=[1])
    if flags & GRID_FLIP1:
        t = t.flip(dims=[2])
    out = torch.zeros((4, R, R, R), dtype=torch.float32)
    a0, a1, a2 = t.shape[1:]
    out[:, :a0, :a1, :a2] = t
    return out, (a0, a1, a2)
