"""Golden 10-step training trace at a head_dim-32 configuration (the only head width the reference backbones use, and the one the HIP
path implements), produced by the REAL reference: `SwinTransformer_MAE3D_New` + torch.optim.AdamW + OneCycleLR + clip_grad_norm_
(run_swin_mae3d.py:588-598,644-669).  Complements G9 (embed_dim 24, head_dim 8: CPU/oracle only).  TEST INFRASTRUCTURE ONLY.
Run: python oracle/gen_golden_trace.py     (needs /root/reference)"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden import _install_shims  # noqa: E402
from oracle.mae3d_oracle import formula_fill_, synthetic_grid  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
KW = dict(patch_size=[4, 4, 4], embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=[4, 4, 4],
          stochastic_depth_prob=0.0, expand_dim=True, resolution=32, masking_prob=0.75)
STEPS = 10


def main():
    assert os.path.isdir(REF), "reference not mounted"
    _install_shims()
    sys.path.insert(0, REF)
    from nerf_mae.model.mae import swin_mae3d as R
    from nerf_mae.model.mae import torch_utils as RU
    torch.set_num_threads(8)
    # embed_dim 32 is not a multiple of 6: the reference's sincos table has 30 channels there and its `pos_embed.copy_` fails (SURVEY
    # 8(c), the swin_b defect).  Same defined deviation as for swin_b: 3 x 10-channel sincos, zero-padded to 32.
    # (the table itself is the oracle's, pinned against the reference's by golden G2 for the widths the reference can build)
    from oracle.mae3d_oracle import sincos_pos_embed_3d

    def padded(embed_dim, grid_size, *a, **k):
        return sincos_pos_embed_3d(embed_dim, grid_size, pad_to=embed_dim)
    RU.get_3d_sincos_pos_embed = padded
    R.get_3d_sincos_pos_embed = padded
    m = R.SwinTransformer_MAE3D_New(**KW)
    formula_fill_(m)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-3)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-4, total_steps=STEPS)
    random.seed(13)
    trace, lrs, betas, gnorm = [], [], [], []
    for step in range(STEPS):
        opt.zero_grad()
        loss, lr_, la_ = m([synthetic_grid((32, 32, 32), 200 + step), synthetic_grid((30, 32, 27), 300 + step)])
        loss.backward()
        gnorm.append(float(torch.nn.utils.clip_grad_norm_(m.parameters(), 0.1)))
        lrs.append(opt.param_groups[0]["lr"])
        betas.append(opt.param_groups[0]["betas"][0])
        opt.step()
        sch.step()
        trace.append([float(loss), float(lr_), float(la_)])
    fin = {n: [p.double().sum().item(), p.double().abs().sum().item()] for n, p in m.named_parameters()}
    np.savez_compressed(os.path.join(OUT, "g13_train_trace_hd32.npz"), trace=np.array(trace, np.float64), lrs=np.array(lrs), beta1=np.array(betas),
                        grad_norm=np.array(gnorm), final_names=np.array(list(fin.keys())), final_sums=np.array(list(fin.values())))
    print("wrote g13_train_trace_hd32.npz; trace:", np.array(trace)[:, 0].round(5).tolist())


if __name__ == "__main__":
    main()
