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


def _worker(rank, world, port, mode, q, comm=None):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nerf_mae_amd.dist import GradReducer, broadcast_parameters
        from nerf_mae_amd.trainer import FusedAdamW, GraphedTrainStep
        m = _build()
        broadcast_parameters(m)
        red = GradReducer(m, comm_dtype=torch.bfloat16 if comm == "bf16" else None)   # bf16 buckets = the default of bench.py at N > 1
        opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-3, max_grad_norm=0.1, eps=1.0)  # eps=1: the update stays linear in tiny gradients, so atomics-order noise is not amplified to +-lr
        grids, bm = _data(rank)
        gavg = None
        if mode == "graph":
            step = GraphedTrainStep(m, opt, 1, reducer=red)
            for _ in range(2):
                step(grids, bm)
        else:
            m._reducer = red
            for it in range(2):
                m.zero_grad()
                m(grids, block_mask=bm)[0].backward()
                red.finish()
                if it == 0:
                    gavg = m._flat_grad.detach().clone()     # the exchanged gradient of the FIRST step (same parameters in every run; before the optimizer clears it)
                opt.step()
        torch.cuda.synchronize()
        flat = m._flat.detach().cpu()
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1]), (both[0] - both[1]).abs().max()
        q.put((rank, "ok", (flat.numpy(), None if gavg is None else gavg.cpu().numpy()) if rank == 0 else None))   # numpy: pickled by value (a tensor would travel as a shared-memory handle)
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("comm", ["fp32", "bf16"])
@pytest.mark.parametrize("mode", ["graph", "eager"])
def test_two_ranks_one_gpu_match_single_process_average(mode, comm):
    """comm = bf16: the gradient buckets travel as bf16 (GradReducer(comm_dtype=torch.bfloat16), what bench.py runs at N > 1; the reference's DDP
    reduces fp32, run_swin_mae3d.py:355-357).  Stated tolerance of the exchanged gradient: each rank's bucket element is rounded to bf16 (2^-9
    relative... see below), the sum once more: bf16 keeps 8 significant bits, round-to-nearest errs by <= 2^-8 = 3.9e-3 relative, once per
    operand and once for the sum -- |mean - exact| <= 8e-3 * (|g0| + |g1|) / 2 per element; the parameters after two steps to 5e-5."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000) + (0 if mode == "graph" else 1) + (2 if comm == "bf16" else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q, comm)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    flat2 = None
    for rank, msg, flat in res:
        assert msg == "ok", f"rank {rank}: {msg}"
        if flat is not None:
            flat2 = torch.from_numpy(flat[0])
            gavg2 = None if flat[1] is None else torch.from_numpy(flat[1])
    # single process: average the two ranks' gradients by hand, same optimizer
    from nerf_mae_amd.trainer import FusedAdamW
    m = _build()
    opt = FusedAdamW(m, lr=1e-3, weight_decay=1e-3, max_grad_norm=0.1, eps=1.0)  # eps=1: the update stays linear in tiny gradients, so atomics-order noise is not amplified to +-lr
    gs_first = None
    for it in range(2):
        gs = []
        for r in range(2):
            grids, bm = _data(r)
            m.zero_grad()
            m(grids, block_mask=bm)[0].backward()
            gs.append(m._flat_grad.clone())
        if it == 0:
            gs_first = [t.clone() for t in gs]
        m._flat_grad.copy_((gs[0] + gs[1]) / 2)
        opt.step()
    torch.cuda.synchronize()
    ref = m._flat.detach().cpu()
    assert ((flat2 - ref).abs().max() / ref.abs().max()).item() < (1e-5 if comm == "fp32" else 5e-5)
    if gavg2 is not None:   # eager mode: the exchanged gradient of the first step, element by element
        g0, g1 = gs_first[0].cpu(), gs_first[1].cpu()
        exact = (g0 + g1) / 2
        bound = (8e-3 if comm == "bf16" else 1e-5) * (g0.abs() + g1.abs()) / 2 + 1e-12 + 1e-5 * exact.abs().max()   # (+ summation-order noise of the fp32 atomics between two runs)
        worst = ((gavg2 - exact).abs() / bound).max().item()
        assert worst <= 1.0, f"exchanged gradient off by {worst:.2f}x the stated bound"


@pytest.mark.parametrize("comm", [None, torch.bfloat16], ids=["fp32-buckets", "bf16-buckets"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 2e-3)], ids=["fp32", "bf16"])
def test_split_graph_step_equals_monolithic_step(dtype, tol, comm):
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
            red = GradReducer(m, comm_dtype=comm)
            red.world, red.active = 2, True    # take the split path ...
            if comm is None:
                red._exchange = lambda lo, hi: None  # ... with an identity exchange (single process)
            else:
                # ... with the bucket casts of the real exchange around an identity collective: flat_grad -> bf16 bucket -> flat_grad (the
                # per-element bf16 rounding, 2^-9 relative, is what the looser tolerance below states)
                from nerf_mae_amd import ops
                red.staging = torch.empty(m._flat_grad.numel(), dtype=comm, device="cuda")

                def _cast_only(lo, hi, red=red, m=m):
                    ops.grad_to_bf16(m._flat_grad[lo:hi], red.staging[lo:hi])
                    ops.grad_from_bf16(red.staging[lo:hi], m._flat_grad[lo:hi], 1.0)
                red._exchange = _cast_only
        step = GraphedTrainStep(m, opt, 2, reducer=red)
        rng = random.Random(11)
        losses = []
        for i in range(3):
            losses.append(step(grids if i == 0 else None, O.draw_block_mask((8, 8, 8), 0.75, rng=rng)).clone())
        torch.cuda.synchronize()
        res.append((torch.stack(losses).cpu(), m._flat.detach().cpu().clone()))
        assert (step._gb1 is not None) == split
    (l0, p0), (l1, p1) = res
    ptol = tol if comm is None else max(tol, 5e-5)    # bf16 buckets: gradients rounded to 2^-9 relative before the (eps = 1) optimizer step
    assert torch.allclose(l0, l1, rtol=max(tol * 10, 0 if comm is None else 1e-4), atol=0), (l0, l1)
    assert ((p0 - p1).abs().max() / p0.abs().max()).item() < ptol


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


def _trace_worker(rank, world, port, comm, q):
    """ten optimizer steps of the g14 run, every rank on the SAME two grids and masks: the mean over ranks then equals the single-process gradient up
    to the bucket dtype, so the curve must reproduce the REAL reference's (golden g14) whatever the bucket dtype"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import random
        from nerf_mae_amd.dist import GradReducer, broadcast_parameters
        from nerf_mae_amd.model import SwinTransformer_MAE3D
        from nerf_mae_amd.trainer import FusedAdamW, OneCycle
        from oracle import mae3d_oracle as O
        from oracle.gen_golden_trace2 import CLIP, EPS, KW, LR, SEED, STEPS, WD, grids
        hip = SwinTransformer_MAE3D(compute_dtype=torch.float32, **KW)
        O.seeded_reference_init_(hip, SEED)
        hip = hip.cuda().train()
        hip.flatten_parameters()
        broadcast_parameters(hip)
        red = GradReducer(hip, comm_dtype=torch.bfloat16 if comm == "bf16" else None)
        hip._reducer = red
        opt = FusedAdamW(hip, lr=LR, weight_decay=WD, max_grad_norm=CLIP, eps=EPS)
        sched = OneCycle(LR, STEPS)
        random.seed(14)
        xs = [t.cuda() for t in grids()]
        trace = []
        for step in range(STEPS):
            lr, b1 = sched.at(step)
            opt.set_hyper(lr=lr, beta1=b1)
            hip.zero_grad()
            loss, l_rgb, l_a = hip(xs)
            loss.backward()
            red.finish()
            opt.step()
            trace.append([loss.item(), l_rgb.item(), l_a.item()])
        q.put((rank, "ok", trace if rank == 0 else None))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc(), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("comm", ["fp32", "bf16"])
def test_two_rank_training_trace_matches_reference_g14(golden, comm):
    """the reference's DDP all-reduces fp32 gradients (run_swin_mae3d.py:355-357); bench.py's default at N > 1 is bf16 buckets.  Ten steps of the
    well-conditioned g14 run on two ranks (gloo, one GPU) must stay within the single-process tolerance of the REAL reference's loss curve (2 % at
    every step, loss / loss_rgb / loss_alpha) with either bucket dtype -- the bar for keeping bf16 the default."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000) + (1 if comm == "bf16" else 0)
    procs = [ctx.Process(target=_trace_worker, args=(r, 2, port, comm, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(60)
    trace = None
    for rank, msg, tr in res:
        assert msg == "ok", f"rank {rank}: {msg}"
        if tr is not None:
            trace = np.array(tr)
    g = golden("g14_train_trace_wellcond.npz")
    print("2 ranks, %s buckets:" % comm, trace[:, 0].round(5).tolist())
    print("reference          :", g["trace"][:, 0].round(5).tolist())
    np.testing.assert_allclose(trace, g["trace"], rtol=2e-2)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (the way the driver starts --gpus 1): bench.py re-executes itself under
    torch.distributed.run (the reference's main() spawns its ranks itself, run_swin_mae3d.py:897-902) and rank 0 prints the one JSON line.
    Dry-run hooks for the one-GPU box: both ranks on device 0 over gloo."""
    import json
    import subprocess
    env = dict(os.environ, NMH_BENCH_BACKEND="gloo", NMH_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--resolution", "32", "--backbone", "swin_t",
                          "--no-cpu-baseline", "--no-kernel-timing"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    assert d["config"].get("collective_ranks_verified") == 2
    assert "comm_ms_exposed" in d["config"]
