"""GPU, world_size 2 on ONE device over gloo (RCCL refuses two ranks on one GPU; the 8-GPU run is the driver's): the control flow of
the data-parallel step -- rank > 0 capture, backward graph -> flat gradient all-reduce -> optimizer graph, and the eager reducer --
must leave both ranks with identical parameters that equal a single-process step on the averaged gradients."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = dict(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8])


def _build():
    from nerf_mae_amd.model import SwinTransformer_MAE3D
    torch.manual_seed(5)
    m = SwinTransformer_MAE3D(patch_size=[4] * 3, window_size=[4] * 3, resolution=32, masking_prob=0.75, stochastic_depth_prob=0.0,
                              compute_dtype=torch.float32, **TINY).cuda()
    m.train()
    m.flatten_parameters()
    return m


def _data(rank):
    from oracle import mae3d_oracle as O
    return [O.synthetic_grid((32, 32, 32), 40 + rank).cuda()], O.draw_block_mask((8, 8, 8), 0.75, rng=__import__("random").Random(9))


def _worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerf_mae_amd.dist import GradReducer, broadcast_parameters
        from nerf_mae_amd.trainer import FusedAdamW, GraphedTrainStep
        m = _build()
        broadcast_parameters(m)
        red = GradReducer(m)
        opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-3, max_grad_norm=0.1, eps=1.0)  # eps=1: the update stays linear in tiny gradients, so atomics-order noise is not amplified to +-lr
        grids, bm = _data(rank)
        if mode == "graph":
            step = GraphedTrainStep(m, opt, 1, reducer=red)
            for _ in range(2):
                step(grids, bm)
        else:
            m._reducer = red
            for _ in range(2):
                m.zero_grad()
                m(grids, block_mask=bm)[0].backward()
                red.finish()
                opt.step()
        torch.cuda.synchronize()
        flat = m._flat.detach().cpu()
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1]), (both[0] - both[1]).abs().max()
        q.put((rank, "ok", flat.numpy() if rank == 0 else None))   # numpy: pickled by value (a tensor would travel as a shared-memory handle)
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["graph", "eager"])
def test_two_ranks_one_gpu_match_single_process_average(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000) + (0 if mode == "graph" else 1)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    flat2 = None
    for rank, msg, flat in res:
        assert msg == "ok", f"rank {rank}: {msg}"
        if flat is not None:
            flat2 = torch.from_numpy(flat)
    # single process: average the two ranks' gradients by hand, same optimizer
    from nerf_mae_amd.trainer import FusedAdamW
    m = _build()
    opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-3, max_grad_norm=0.1, eps=1.0)  # eps=1: the update stays linear in tiny gradients, so atomics-order noise is not amplified to +-lr
    for _ in range(2):
        gs = []
        for r in range(2):
            grids, bm = _data(r)
            m.zero_grad()
            m(grids, block_mask=bm)[0].backward()
            gs.append(m._flat_grad.clone())
        m._flat_grad.copy_((gs[0] + gs[1]) / 2)
        opt.step()
    torch.cuda.synchronize()
    ref = m._flat.detach().cpu()
    assert ((flat2 - ref).abs().max() / ref.abs().max()).item() < 1e-5


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 2e-3)], ids=["fp32", "bf16"])
def test_split_graph_step_equals_monolithic_step(dtype, tol):
    """the data-parallel graph step cuts the backward into three captured pieces (decoder | stages 3,2 | stages 1,0,embed) with the
    gradient exchange between the replays; with a no-op exchange it must reproduce the single-graph step (same masks, same inputs,
    stochastic depth off): equal losses and equal parameters after three optimizer steps"""
    import random
    from nerf_mae_amd.dist import GradReducer
    from nerf_mae_amd.model import SwinTransformer_MAE3D
    from nerf_mae_amd.trainer import FusedAdamW, GraphedTrainStep
    from oracle import mae3d_oracle as O
    grids = [O.synthetic_grid((32, 32, 32), 3).cuda(), O.synthetic_grid((32, 28, 30), 4).cuda()]
    res = []
    for split in (False, True):
        torch.manual_seed(5)
        m = SwinTransformer_MAE3D(patch_size=[4] * 3, window_size=[4] * 3, resolution=32, masking_prob=0.75, stochastic_depth_prob=0.0,
                                  compute_dtype=dtype, **TINY).cuda()
        m.train()
        m.flatten_parameters()
        opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-3, max_grad_norm=0.1, eps=1.0)
        red = None
        if split:
            red = GradReducer(m)
            red.world, red.active = 2, True    # take the split path ...
            red._exchange = lambda lo, hi: None  # ... with an identity exchange (single process)
        step = GraphedTrainStep(m, opt, 2, reducer=red)
        rng = random.Random(11)
        losses = []
        for i in range(3):
            losses.append(step(grids if i == 0 else None, O.draw_block_mask((8, 8, 8), 0.75, rng=rng)).clone())
        torch.cuda.synchronize()
        res.append((torch.stack(losses).cpu(), m._flat.detach().cpu().clone()))
        assert (step._gb1 is not None) == split
    (l0, p0), (l1, p1) = res
    assert torch.allclose(l0, l1, rtol=tol * 10, atol=0), (l0, l1)
    assert ((p0 - p1).abs().max() / p0.abs().max()).item() < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_segment_triggers_fire_after_every_weight_gradient_of_the_segment(dtype):
    """eager data-parallel mode over RCCL launches a segment's all-reduce from an autograd trigger: every weight-gradient launch of that
    segment -- including the ones queued for a later grouped / forked issue -- must already be in the stream.  A recording stand-in for the
    reducer snapshots the flat gradient at each trigger; the slice of the segment must equal the final gradient."""
    sys.path.insert(0, ROOT)
    from nerf_mae_amd.dist import GradReducer, _Trigger
    from nerf_mae_amd.model import SwinTransformer_MAE3D
    torch.manual_seed(5)
    m = SwinTransformer_MAE3D(patch_size=[4] * 3, window_size=[4] * 3, resolution=32, masking_prob=0.75, stochastic_depth_prob=0.0,
                              compute_dtype=dtype, **TINY).cuda()
    m.train()
    m.flatten_parameters()
    real = GradReducer(m)          # world 1: only its segment bounds are used

    class Recorder:
        bounds, nseg, snaps = real.bounds, real.nseg, {}
        chunk_stage, chunk_groups, seg_stage, seg_decoder = real.chunk_stage, real.chunk_groups, real.seg_stage, real.seg_decoder

        def trigger(self, x, seg):
            return _Trigger.apply(x, self, seg)

        def launch(self, seg):
            torch.cuda.synchronize()
            self.snaps[seg] = m._flat_grad[self.bounds[seg]:self.bounds[seg + 1]].clone()

    rec = Recorder()
    m._reducer = rec
    grids, bm = _data(0)
    m.zero_grad()
    m(grids, block_mask=bm)[0].backward()
    torch.cuda.synchronize()
    assert len(rec.snaps) == real.nseg - 1, "every stage / block-group / decoder trigger fires (the embed range is finish()'s)"
    for seg, snap in rec.snaps.items():
        final = m._flat_grad[rec.bounds[seg]:rec.bounds[seg + 1]]
        assert torch.equal(snap, final), (seg, (snap - final).abs().max().item())
