"""CPU (-m "not gpu"): the C-ABI library loads and exports every symbol include/*.h declares (no compute calls), and the
host-side logic (tables, masking RNG, scheduler, sharding, state-dict contract, loud failure without a GPU)."""
import os
import random

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_library_loads_and_exports_every_declared_symbol():
    from nerf_mae_amd._lib import HEADER, LIB_PATH, lib, parse_header
    assert os.path.exists(LIB_PATH), "run `python __graft_entry__.py` (build()) first"
    sigs = parse_header(HEADER)
    assert len(sigs) >= 28 and "nmh_window_attn_bwd" in sigs and "nmh_conv3d_k3_wgrad" in sigs
    L = lib()
    for name, (ret, args) in sigs.items():
        assert hasattr(L.cdll, name), f"{name} declared in nerfmae_hip.h but not exported"
    assert L.call("nmh_version") >= 100
    assert L.cdll.nmh_error_string(-2).decode().startswith("nmh:")
    # every NMH_API line parsed (the regex must not silently drop declarations)
    n_decl = sum(1 for line in open(HEADER) if line.startswith("NMH_API"))
    assert n_decl == len(sigs)


def test_no_cpu_fallback_product_fails_loudly_without_device():
    from nerf_mae_amd import ops
    from nerf_mae_amd.model import build_model
    with pytest.raises(RuntimeError):
        ops.gemm_nt(torch.zeros(8, 8), torch.zeros(8, 8))
    m = build_model("swin_t", resolution=32)
    with pytest.raises(RuntimeError):
        m([torch.rand(4, 32, 32, 32)])
    # the product package never imports the oracle
    import sys
    for f in os.listdir(os.path.join(ROOT, "nerf-mae_amd")):
        if f.endswith(".py"):
            src = open(os.path.join(ROOT, "nerf-mae_amd", f)).read()
            assert "import oracle" not in src and "from oracle" not in src, f


def test_tables_and_mask_rng_match_reference_golden(golden):
    from nerf_mae_amd.model import draw_block_mask, relative_position_index, sincos_pos_embed_3d
    g2, g5 = golden("g2_tables.npz"), golden("g5_masks.npz")
    assert np.array_equal(relative_position_index(4).numpy().astype(np.int16), g2["rel_index"])
    np.testing.assert_allclose(sincos_pos_embed_3d(24, 8).astype(np.float32), g2["pos_embed_24_8"], atol=1e-6)
    big = sincos_pos_embed_3d(96, 40).astype(np.float32)
    np.testing.assert_allclose(big[0, ::13, ::11, ::7, :], g2["pos_embed_96_40_samples"], atol=1e-6)
    for key, gsz, seed, p in [("blocks_40_seed1234", 40, 1234, 0.75), ("blocks_8_seed1234", 8, 1234, 0.75), ("blocks_40_seed7_p50", 40, 7, 0.5)]:
        random.seed(seed)
        assert np.array_equal(draw_block_mask((gsz,) * 3, p)[::4, ::4, ::4].numpy(), g5[key]), key
    # swin_b deviation: 3 x 42 channels + 2 zero pads
    pe = sincos_pos_embed_3d(128, 4)
    assert pe.shape == (1, 4, 4, 4, 128) and np.all(pe[..., 126:] == 0)


def test_state_dict_contract_and_strict_interchange_with_oracle():
    from nerf_mae_amd.model import SwinTransformer_MAE3D, SwinTransformer_MAE3D_New, build_model
    from oracle import mae3d_oracle as O
    assert SwinTransformer_MAE3D is SwinTransformer_MAE3D_New
    for name, nkeys in [("swin_t", 215), ("swin_s", 383)]:
        m, o = build_model(name, resolution=32), O.build_oracle(name, resolution=32)
        sd, so = m.state_dict(), o.state_dict()
        assert len(sd) == nkeys and set(sd) == set(so)
        assert all(sd[k].shape == so[k].shape and sd[k].dtype == so[k].dtype for k in sd)
        assert not m.load_state_dict(so, strict=True).missing_keys
        assert not o.load_state_dict(sd, strict=True).missing_keys
    m = build_model("swin_s", resolution=160)
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 70040938
    assert m.pos_embed.requires_grad is False and m.pos_embed.shape == (1, 40, 40, 40, 96)
    # nerf_rpn deletes the decoder half after a strict load (feature_extractor.py:1158-1163)
    del m.decoder4, m.decoder3, m.decoder2, m.decoder1, m.out, m.mask_token
    assert all(k.startswith(("patch_partition", "stages", "pos_embed")) for k in m.state_dict())
    with pytest.raises(ValueError):
        SwinTransformer_MAE3D(patch_size=[4] * 3, embed_dim=128, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24], window_size=[4] * 3)


def test_flat_buffers_and_packer_layout_on_cpu():
    """flatten_parameters keeps values, makes params/grads views of two flat buffers (mask_token first: its gradient completes last)"""
    from nerf_mae_amd.model import build_model
    m = build_model("swin_t", resolution=32, compute_dtype=torch.float32)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    flat, fg = m.flatten_parameters(torch.device("cpu"))
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k
    assert m.mask_token.data_ptr() == flat.data_ptr()
    for p in m.parameters():
        if p.requires_grad:
            off = m._offsets[id(p)]
            assert p.data_ptr() == flat.data_ptr() + 4 * off and p.grad.data_ptr() == fg.data_ptr() + 4 * off and off % 4 == 0
    P = m._packer
    assert P["decoder1.c1.w"].numel() == 48 * 27 * 48 and P["s2.1.qkv.wT"].numel() == 3 * 384 * 384
    assert P.descs.numel() == 40 * len(P.items)


def test_onecycle_matches_torch_onecyclelr():
    from nerf_mae_amd.trainer import OneCycle
    p = torch.nn.Parameter(torch.zeros(1))
    for total in (10, 57, 2000):
        opt = torch.optim.AdamW([p], lr=1e-4)
        sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-4, total_steps=total)
        mine = OneCycle(1e-4, total)
        for step in range(total):
            lr, b1 = mine.at(step)
            assert abs(lr - opt.param_groups[0]["lr"]) < 1e-12 and abs(b1 - opt.param_groups[0]["betas"][0]) < 1e-9, (total, step)
            opt.step()
            sch.step()


def test_shard_indices_match_distributed_sampler():
    from torch.utils.data import DistributedSampler
    from nerf_mae_amd.dist import shard_indices
    ds = list(range(37))
    for world in (1, 2, 8):
        for rank in range(world):
            s = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=0)
            for epoch in (0, 3):
                s.set_epoch(epoch)
                assert list(iter(s)) == shard_indices(len(ds), rank, world, epoch)


def test_window_geometry():
    from nerf_mae_amd.ops import WinGeom
    g = WinGeom(2, 5, 10, 2, [2, 2, 2])
    assert g.P == [8, 12, 4] and g.shift == [2, 2, 0] and g.rows == 2 * 8 * 12 * 4 and g.tokens == 200
    assert list(g.carr) == [2, 5, 10, 2, 8, 12, 4, 2, 2, 0]


def test_block_bits_roundtrip_and_rng_stream():
    """draw_block_bits consumes the python RNG exactly like draw_block_mask (the reference's window_masking_3d raster, swin_mae3d.py:
    1366-1373) and block_bits_of_mask inverts the block fill; non-block-structured masks are rejected (-> the upload fallback)"""
    import random
    import numpy as np
    from nerf_mae_amd.model import block_bits_of_mask, draw_block_bits, draw_block_mask
    for g in (8, 10, 40):
        m = draw_block_mask((g, g, g), 0.75, rng=random.Random(7)).numpy()
        bits = draw_block_bits((g, g, g), 0.75, rng=random.Random(7))
        back = block_bits_of_mask(m)
        assert back is not None and np.array_equal(back, bits)
    bad = draw_block_mask((8, 8, 8), 0.5, rng=random.Random(1)).numpy().copy()
    bad[1, 2, 3] ^= 1
    assert block_bits_of_mask(bad) is None


def test_capi_rejects_null_required_pointers_without_touching_the_device():
    """boundary contract (include/nerfmae_hip.h): a NULL in a required pointer argument returns the argument-error code -4 -- no launch,
    no device fault -- so the check runs on a box without a GPU"""
    import ctypes
    from nerf_mae_amd._lib import lib
    L = lib().cdll
    null = ctypes.c_void_p(None)
    assert L.nmh_gemm_nt(1, null, 96, null, 96, 64, 96, 96, null, 96, null, 0, null, null, null, 1, 0, null) == -4
    assert L.nmh_conv3d_k3(1, null, null, null, 1, 8, 8, 8, 48, 48, 0, null, 0, null) == -4
    assert L.nmh_conv3d_k3_c48mb(null, null, null, 1, 8, 8, 8, 96, 96, 0, null) == -4
    assert L.nmh_gemm_tn_grouped(1, null, 3, null, 0, null) == -4
    assert L.nmh_conv3d_k3_c64(null, null, null, 1, 8, 8, 8, 64, 64, 0, null, null, null) == -4
    assert L.nmh_set_prezeroed_arena(null, 4096) == -4 and L.nmh_set_prezeroed_arena(null, -1) == -4
    assert L.nmh_set_prezeroed_arena(null, 0) == 0      # bytes = 0 removes the arena
    assert L.nmh_error_string(-4).decode().startswith("nmh:")
