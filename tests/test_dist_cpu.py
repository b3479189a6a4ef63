"""CPU, gloo, world_size 2: the data-parallel gradient exchange (segment all-reduce launched from backward triggers,
AVG semantics == DDP gradient averaging, initial parameter broadcast)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerf_mae_amd.dist import GradReducer, broadcast_parameters
        from nerf_mae_amd.model import build_model
        torch.manual_seed(100 + rank)  # different init per rank -> broadcast must equalise
        m = build_model("swin_t", resolution=32, compute_dtype=torch.float32)
        m.flatten_parameters(torch.device("cpu"))
        broadcast_parameters(m)
        ref = m._flat.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, m._flat)
        red = GradReducer(m)
        # swin_t: stage 2 has 6 blocks -> three groups of two; segments embed | stage0 | stage1 | 3 x stage-2 groups | stage3 | decoders
        assert red.nseg == 8 and red.nchunks == 3 and red.bounds[0] == 0 and red.bounds[-1] == m._flat_grad.numel()
        assert red.bounds == sorted(red.bounds) and len(set(red.bounds)) == len(red.bounds)
        assert [red.seg_stage(0), red.seg_stage(1), red.seg_stage(2, 0), red.seg_stage(2, 2), red.seg_stage(3), red.seg_decoder()] == [1, 2, 3, 5, 6, 7]
        assert red.bounds[red.seg_stage(2, 1)] == m._offsets[id(next(red.chunk_groups[1][0].parameters()))]
        assert red.stage_bounds[3] == red.bounds[3] and red.stage_bounds[4] == red.bounds[6] and len(red.stage_bounds) == 7
        # segment 0 holds mask_token + patch embed; last segment starts at decoder4
        assert red.bounds[1] == m._offsets[id(next(m.stages[0].parameters()))]
        # fake backward: rank-specific gradients, triggers fire in backward order (decoder, stage3..0), finish() does the rest
        g = m._flat_grad
        g.copy_(torch.arange(g.numel(), dtype=torch.float32) * (rank + 1) * 1e-3)
        expect = torch.arange(g.numel(), dtype=torch.float32) * 1e-3 * (sum(range(1, world + 1)) / world)
        x = torch.ones(3, requires_grad=True)
        y = x
        order = []
        orig = red.launch
        red.launch = lambda seg: (order.append(seg), orig(seg))[1]
        for seg in (1, 2, 3, 4, 5, 6, 7):       # forward order: stage0, stage1, the three stage-2 groups, stage3 inputs, then decoder input
            y = red.trigger(y * 1.0, seg)
        y.sum().backward()
        red.finish()
        assert order == [7, 6, 5, 4, 3, 2, 1, 0], order
        assert torch.allclose(g, expect, rtol=1e-6), (g - expect).abs().max()
        assert torch.allclose(x.grad, torch.ones(3))
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_grad_reducer_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
