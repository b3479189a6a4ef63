"""CPU fp32 restatement of the two dense-prediction heads the reference fine-tunes on top of the MAE-pretrained encoder + decoder
(SURVEY 8(f) rank 4):  `SwinTransformer_VoxelSR_Pretrained_Skip` (nerf_rpn/model/feature_extractor.py:1898-2244) and
`SwinTransformer_VoxelSemantics_Pretrained_Skip` (:2521-2848).  TEST INFRASTRUCTURE ONLY: imported by tests/ (the checker of the HIP
path) and by oracle/gen_golden_heads.py, which pins it against the REAL reference classes (golden g15)."""
from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .mae3d_oracle import OutBlock3D, ResBlock3D, UpBlock3D, build_oracle, pad_grid


class BasicBlock3D(nn.Module):
    """UnetrBasicBlock(res_block=True) (unetr_block.py:323-370): a UnetResBlock under the attribute `layer`."""

    def __init__(self, cin, cout):
        super().__init__()
        self.layer = ResBlock3D(cin, cout)

    def forward(self, x):
        return self.layer(x)


class _HeadBase(nn.Module):
    def __init__(self, resolution: int, backbone_type: str = "swin_s", **kw):
        super().__init__()
        base = build_oracle(backbone_type, resolution=resolution, stochastic_depth_prob=kw.pop("stochastic_depth_prob", 0.1), **kw)
        del base.decoder1, base.out, base.mask_token        # feature_extractor.py:2006-2008 / 2609-2611
        self.base = base
        E = base.embed_dim
        self.input_resolution = resolution
        self.encoder1 = BasicBlock3D(4, E // 2)                # :2019-2025
        self.decoder1 = UpBlock3D(E, E // 2, 4, use_skip=True)  # :2027-2034

    def _dec0(self, x: List[Tensor]) -> Tensor:
        """forward up to the last decoder level (feature_extractor.py:2199-2222 / 2797-2817)"""
        xb = torch.cat([pad_grid(t, self.input_resolution)[0] for t in x])
        enc1 = self.encoder1(xb)
        b = self.base
        t = b.patch_partition(xb)
        t = t + b.pos_embed
        feats = []
        for st in b.stages:
            t = st(t)
            feats.append(t.permute(0, 4, 1, 2, 3).contiguous())
        d3 = b.decoder4(feats[3], feats[2])
        d2 = b.decoder3(d3, feats[1])
        d1 = b.decoder2(d2, feats[0])
        return self.decoder1(d1, enc1)


class VoxelSROracle(_HeadBase):
    """voxel super-resolution: dec0 -> nn.Upsample(scale_factor = out/160, nearest) -> 1x1x1 conv to 4 channels (:2036,2047,2224-2229);
    loss = masked RGB MSE against the grid at the output resolution (:2133-2160)."""

    def __init__(self, resolution: int = 160, out_resolution: int = 256, **kw):
        super().__init__(resolution, **kw)
        self.output_resolution = out_resolution
        self.scale = {256: 1.6, 384: 2.4}.get(out_resolution, out_resolution / resolution)
        self.voxel_out = OutBlock3D(self.base.embed_dim // 2, 4)

    def forward(self, x: List[Tensor]) -> Tensor:
        return self.voxel_out(F.interpolate(self._dec0(x), scale_factor=self.scale, mode="nearest"))

    def forward_loss(self, x: List[Tensor], pred: Tensor) -> Tensor:
        tgt = torch.cat([pad_grid(t, self.output_resolution)[0] for t in x]).permute(0, 2, 3, 4, 1)
        p = pred.permute(0, 2, 3, 4, 1)
        m = (tgt[..., 3:] > 0.01).int()
        return (((p[..., :3] - tgt[..., :3]) ** 2) * m).sum() / m.sum()


class VoxelSemanticsOracle(_HeadBase):
    """voxel semantics: dec0 -> 1x1x1 conv to `out_channels` class logits (:2641-2643,2818); loss = nn.CrossEntropyLoss(weight) over ALL
    voxels of (logits * mask, labels * mask), mask = labels > 0 (metrics.py:540-553, feature_extractor.py:2700-2722); the mIoU figure
    (metrics.py:194-245) is a metric, not part of the loss."""

    def __init__(self, resolution: int = 160, out_channels: int = 19, class_weights: Optional[Tensor] = None, **kw):
        super().__init__(resolution, **kw)
        self.out_channels = out_channels
        self.class_weights = class_weights
        self.sem_out = OutBlock3D(self.base.embed_dim // 2, out_channels)

    def forward(self, x: List[Tensor]) -> Tensor:
        return self.sem_out(self._dec0(x))

    def forward_loss(self, labels: List[Tensor], pred: Tensor):
        K = self.out_channels
        tgt = torch.cat([pad_grid(t, self.input_resolution)[0] for t in labels]).permute(0, 2, 3, 4, 1).long()   # (B,R,R,R,1)
        p = pred.permute(0, 2, 3, 4, 1)
        mask = (tgt > 0).long()
        loss = F.cross_entropy((p * mask).reshape(-1, K), (tgt * mask).squeeze(-1).reshape(-1), weight=self.class_weights)
        # mIoULoss_new (metrics.py:194-245): soft IoU per (sample, class >= 1) over the masked voxels
        prob = F.softmax(p, dim=-1) * mask
        onehot = torch.zeros_like(prob).scatter_(-1, tgt * mask, 1.0)
        N = p.shape[0]
        inter = (prob[..., 1:] * onehot[..., 1:]).reshape(N, -1, K - 1).sum(1)
        union = (prob[..., 1:] + onehot[..., 1:] - prob[..., 1:] * onehot[..., 1:]).reshape(N, -1, K - 1).sum(1)
        iou = inter / (union + 1e-8)
        return loss, loss, iou.mean()
