#!/usr/bin/env python
"""bench.py -- NeRF-MAE hot-path benchmark on MI355X (contract: see the task brief / DESIGN.md "Measurement").

One "step" = zero_grad + forward + backward (+ gradient all-reduce when N>1) + clip + AdamW on one batch of synthetic
160^3 RGB-sigma grids already resident in HBM.  Metric: voxel-grids/s (whole job), swin_s, 160^3, bf16 (BASELINE.json).

Workload of `value`: BASELINE.json's headline config read literally -- swin_s, GLOBAL batch 8 x (4 x 160^3), DP = N: at N = 1 the
whole batch of 8 grids runs on the one GPU, at N = 8 every GPU steps on 1 grid ("scaling": "strong").  `--batch-per-gpu B` fixes the
per-GPU batch instead (weak scaling).  The same JSON line carries, under config.sweep, the 1-, 4- and 8-grids-per-GPU figures of this
GPU (the headline's weak-scaling share, the reference's training recipe of train_mae3d.sh, and the whole headline batch).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: the 3x3x3 48->48 decoder conv, MFMA-bound; plus the three slowest
HBM-bound kernels against 8 TB/s) and `cpu_baseline` (the CPU oracle timed on the host cores on a bounded sample).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import hashlib
import json
import os
import random
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def fwd_flops_per_grid(cfg, R):
    """SURVEY 8(d) analytic model (2*MAC, GEMM-like ops on real tokens)."""
    C, depths = cfg["embed_dim"], cfg["depths"]
    g = R // 4
    fl = 2.0 * g ** 3 * 256 * C
    s = g
    for i, d in enumerate(depths):
        c = C * 2 ** i
        if i > 0:
            s = (s + 1) // 2
            fl += 2.0 * s ** 3 * (4 * c) * c  # merge: 8*(c/2) -> c
        T = s ** 3
        fl += d * (2.0 * T * c * 3 * c + 2.0 * T * c * c + 4.0 * T * 64 * c + 16.0 * T * c * c)
    E = C
    v = s
    for cin, cout, k, skip in ((8 * E, 4 * E, 2, True), (4 * E, 2 * E, 2, True), (2 * E, E, 2, True), (E, E // 2, 4, False)):
        V = (v * k) ** 3
        fl += 2.0 * V * cin * cout                                   # transpose conv
        cc = 2 * cout if skip else cout
        fl += 2.0 * V * 27 * cc * cout + 2.0 * V * 27 * cout * cout  # two 3^3 convs
        if skip:
            fl += 2.0 * V * cc * cout                                # 1x1 residual conv
        v *= k
    fl += 2.0 * R ** 3 * (E // 2) * 4
    return fl


def _self_launch(n: int) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: re-run the same command line as N ranks of one node."""
    import socket
    import subprocess
    with socket.socket() as sk:       # a free port for the rendezvous (the reference picks a random one, run_swin_mae3d.py:890-893)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL / CUDA-IPC across processes needs it on this driver
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--backbone", default="swin_s")
    ap.add_argument("--resolution", type=int, default=160)
    ap.add_argument("--global-batch", type=int, default=8,
                    help="grids per step over ALL GPUs (strong scaling): BASELINE configs[2], the headline, is global batch 8")
    ap.add_argument("--batch-per-gpu", type=int, default=0,
                    help="fix the grids per GPU per step instead (weak scaling): 1 = the headline's share at DP=8, 4 = the reference's training "
                         "recipe (train_mae3d.sh: batch 32 on 8 GPUs; BASELINE configs[1])")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the 1/4/8 grids-per-GPU sweep legs (config.sweep)")
    ap.add_argument("--e2e", action="store_true", help="also time Trainer.fit on host-resident synthetic scenes (uint8 and fp32 storage)")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python each step instead of replaying the captured HIP graph")
    args = ap.parse_args()

    from nerf_mae_amd import ops
    from nerf_mae_amd.dist import GradReducer, broadcast_parameters
    from nerf_mae_amd.model import SWIN_CONFIGS, build_model, draw_block_mask
    from nerf_mae_amd.trainer import FusedAdamW, GraphedTrainStep, OneCycle
    from nerf_mae_amd import data

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: spawn the ranks ourselves, as the reference's main() does (run_swin_mae3d.py:897-902 mp.spawn) -- here by
        # re-executing this command under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1); rank 0's JSON line is the output
        return _self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    # dry-run hooks for a 1-GPU box (control flow of the N > 1 path: rank > 0 capture, split graphs, optimizer graph): every rank
    # on device 0 and gloo instead of RCCL (which refuses two ranks on one GPU).  Never set by the driver.
    backend = os.environ.get("NMH_BENCH_BACKEND", "nccl")
    if os.environ.get("NMH_BENCH_SHARE_GPU", "0") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rccl_ranks = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        # the collective really spans N ranks: all-reduce of a one-hot rank vector must come back all ones
        assert dist.get_world_size() == args.gpus
        onehot = torch.zeros(world, device=dev)
        onehot[rank] = 1.0
        dist.all_reduce(onehot)
        torch.cuda.synchronize()
        assert bool((onehot == 1).all()), f"rank-count check failed: {onehot.tolist()}"
        rccl_ranks = int(onehot.sum().item())

    R = args.resolution
    if args.batch_per_gpu > 0:
        Bg, scaling = args.batch_per_gpu, "weak"
    else:
        assert args.global_batch % world == 0, "--global-batch must be a multiple of --gpus"
        Bg, scaling = args.global_batch // world, "strong"
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    random.seed(0)
    model = build_model(args.backbone, resolution=R, masking_prob=0.75, stochastic_depth_prob=0.1, compute_dtype=dtype).to(dev)
    model.train()
    model.flatten_parameters()
    broadcast_parameters(model)
    reducer = GradReducer(model, comm_dtype=torch.bfloat16 if os.environ.get("NMH_COMM_BF16", "1") == "1" else None) if world > 1 else None
    model._reducer = reducer
    opt = FusedAdamW(model, lr=1e-4, weight_decay=1e-3, max_grad_norm=0.1)
    sched = OneCycle(1e-4, 1000)
    step_no = [0]

    # synthetic inputs: valid extents cycle through {160^3, 160x132x96, 120x160x144}, resident in HBM
    exts = [(R, R, R), (R, int(R * 0.825), int(R * 0.6)), (int(R * 0.75), R, int(R * 0.9))]
    # N = 1: the 1 / 4 / 8 grids-per-GPU figures of this GPU.  N > 1: BOTH scaling legs in the one line -- `value` is the strong leg (global batch 8
    # split over the ranks); config.sweep adds the weak legs at 8 and 1 grids per GPU, each with its own comm_ms_exposed
    sweep_sizes = [] if (args.no_sweep or args.eager) else [b for b in ((1, 4, 8) if world == 1 else (8, 1)) if b != Bg]
    nmax = max([Bg] + sweep_sizes)
    # stored-format scenes (W,L,H,4 with raw density) go through the product input pipeline (density->alpha, layout, padding on the GPU)
    scenes = [data.synthetic_scene(exts[(rank * nmax + i) % 3], seed=rank * 131 + i) for i in range(nmax)]
    xb0, ext0 = data.GridBatcher(R, dev, normalize_density=True)(scenes, flags=[0] * nmax)
    grids_all = [xb0[i, :, :e[0], :e[1], :e[2]].contiguous() for i, e in enumerate(ext0.tolist())]
    del xb0
    mask_rng = random.Random(1000 + rank)
    g = R // 4

    def barrier():
        if world > 1:
            if backend == "nccl":
                dist.barrier(device_ids=[local])
            else:
                torch.cuda.synchronize()
                dist.barrier()
        torch.cuda.synchronize()

    comm_events = {}

    def run_leg(nb, steps, warmup, use_reducer=True, comm_timing_key=None):
        """`steps` timed steps at nb grids per GPU -> (seconds [max over ranks], last loss).  comm_timing_key: HIP events around every gradient exchange
        (comm stream) and at the join (compute stream) during the timed steps -> comm_events[key] (per-range all-reduce ms, exposed ms from timestamps)"""
        grids = grids_all[:nb]
        graphed = None
        red = reducer if use_reducer else None
        if not args.eager:
            graphed = GraphedTrainStep(model, opt, nb, reducer=red)
            graphed(grids, draw_block_mask((g, g, g), 0.75, rng=mask_rng))  # loads the static batch, captures (lr is 0 until update_hyper)
        else:
            model._reducer = red

        def step():
            lr, b1 = sched.at(step_no[0])
            step_no[0] += 1
            opt.set_hyper(lr=lr, beta1=b1)
            bm = draw_block_mask((g, g, g), 0.75, rng=mask_rng)   # per-step python-random mask, as the reference
            if graphed is not None:
                return graphed(None, bm)[0]
            model.zero_grad()
            loss, l_rgb, l_a = model(grids, block_mask=bm)
            loss.backward()
            if red is not None:
                red.finish()
            opt.step()
            return loss

        for _ in range(warmup):
            loss = step()
        barrier()
        if comm_timing_key is not None and red is not None:
            red.timing_begin()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        barrier()
        dt = time.perf_counter() - t0
        if comm_timing_key is not None and red is not None:
            comm_events[comm_timing_key] = red.timing_report(steps)
        if world > 1:
            t = torch.tensor([dt], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        lv = loss.item()
        del graphed
        torch.cuda.empty_cache()
        return dt, lv

    dt, loss_val = run_leg(Bg, args.steps, args.warmup)
    grids_per_s = args.steps * Bg * world / dt
    cfg = SWIN_CONFIGS[args.backbone]
    flops_fb = 3.0 * fwd_flops_per_grid(cfg, R)
    frac = lambda gps_per_gpu: round(gps_per_gpu * flops_fb / (PEAK_BF16_TFLOPS * 1e12), 4)  # noqa: E731

    comm_exposed = None
    if world > 1:
        # the same K steps without the gradient exchange (every rank steps on its local gradients): the difference is the part of the
        # all-reduce that the backward did not hide
        dt_local, _ = run_leg(Bg, args.steps, args.warmup, use_reducer=False)
        comm_exposed = round(1e3 * (dt - dt_local) / args.steps, 3)
        broadcast_parameters(model)
        # a third, short leg with HIP events on the comm stream: per-range all-reduce times and the exposed part from stream timestamps
        run_leg(Bg, max(3, min(args.steps, 5)), 2, comm_timing_key="%d_grids_per_gpu" % Bg)

    sweep = {}
    for nb in sweep_sizes:
        ks = max(3, min(args.steps, 10))
        d2, _ = run_leg(nb, ks, 3)
        gps = ks * nb * world / d2
        sweep["%d_grids_per_gpu" % nb] = {"grids_per_s": round(gps, 3), "ms_per_step": round(1e3 * d2 / ks, 3), "whole_step_mfma_frac": frac(gps / world)}
        if world > 1:   # weak-scaling leg: global batch nb * N; its exposed communication as for the headline leg
            d3, _ = run_leg(nb, ks, 3, use_reducer=False)
            broadcast_parameters(model)
            sweep["%d_grids_per_gpu" % nb].update({"scaling": "weak", "global_batch": nb * world, "comm_ms_exposed": round(1e3 * (d2 - d3) / ks, 3)})
            run_leg(nb, 3, 2, comm_timing_key="%d_grids_per_gpu" % nb)
    sweep["%d_grids_per_gpu" % Bg] = {"grids_per_s": round(grids_per_s, 3), "ms_per_step": round(1e3 * dt / args.steps, 3),
                                      "whole_step_mfma_frac": frac(grids_per_s / world)}
    if world > 1:
        sweep["%d_grids_per_gpu" % Bg].update({"scaling": scaling, "global_batch": Bg * world, "comm_ms_exposed": comm_exposed})

    # per-kernel durations with HIP events on the launch stream: a few extra eager steps on the same model/data right after
    # the timed region (graph replays cannot carry per-launch events; the kernels, shapes and data are identical).  The side stream
    # is off for these steps so that an event pair brackets exactly one kernel (in the replayed step the decoder weight pack and the
    # weight-gradient GEMMs overlap other kernels; rocprofv3 of the replay shows the same per-launch time, profiles/)
    prof = None
    ksteps = min(3, args.steps)
    if not args.no_kernel_timing and rank == 0:
        side_was, ops.side_stream.enabled = ops.side_stream.enabled, False
        red_was, model._reducer = model._reducer, None
        ops.PROFILE = {}
        for i in range(ksteps + 1):
            if i == 1:
                torch.cuda.synchronize()
                ops.PROFILE = {}       # first eager step after the replays: lazy allocations, not timed
            model.zero_grad()
            l3 = model(grids_all[:Bg], block_mask=draw_block_mask((g, g, g), 0.75, rng=mask_rng))
            l3[0].backward()
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        ops.side_stream.enabled = side_was
        model._reducer = red_was
    barrier()

    headline = (args.backbone == "swin_s" and R == 160 and args.dtype == "bf16")
    out = {
        "metric": "voxel-grids/sec (fwd+bwd+optimizer) %s %d^3 %s" % (args.backbone, R, args.dtype),
        "value": round(grids_per_s, 4), "unit": "grids/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "%s MAE pretraining step, global batch %d = %d grid(s)/GPU of 4x%d^3 RGB-sigma, mask_ratio 0.75, stochastic depth 0.1, AdamW+clip, DP=%d%s"
                               % (args.backbone, Bg * world, Bg, R, world,
                                  " (BASELINE configs[2], the headline, read literally)" if headline and Bg * world == 8 else ""),
                   "global_batch": Bg * world, "grids_per_gpu": Bg, "resolution": R, "parallelism": "dp%d" % world,
                   "launch": "eager" if args.eager else "hipgraph",
                   "algorithmic_tflop_per_grid_fwd_bwd": round(flops_fb / 1e12, 3),
                   "whole_step_mfma_frac": frac(grids_per_s / world), "final_loss": round(loss_val, 5),
                   "sweep": sweep},
    }
    if world > 1:
        out["config"]["collective_ranks_verified"] = rccl_ranks
        out["config"]["comm_ms_exposed"] = comm_exposed
        out["config"]["grad_comm_dtype"] = "bf16" if (reducer is not None and reducer.comm_dtype == torch.bfloat16) else "fp32"
        # one record for the scaling run: `value` / config.sweep[<Bg>] = the strong leg (global batch 8 over N ranks), config.sweep["8_grids_per_gpu"] /
        # ["1_grids_per_gpu"] = the weak legs; comm_ms_exposed = step time with minus without the exchange (same K steps), comm_events = HIP-event
        # timestamps of a short extra leg: every range's all-reduce on the comm stream and what the compute stream waited for at the join
        out["config"]["comm_ms_exposed_method"] = "K timed steps with the gradient exchange minus the same K steps on local gradients"
        out["config"]["comm_events"] = comm_events
        out["config"]["legs"] = {"strong": "%d_grids_per_gpu" % Bg if scaling == "strong" else None,
                                 "weak": [k for k, v in sweep.items() if v.get("scaling") == "weak"]}

    if rank == 0 and prof:
        # dominant kernel: implicit-GEMM 3x3x3 conv at R^3 with Cin=Cout=E/2 (decoder1 fwd + dgrad launches share one kernel)
        E2 = cfg["embed_dim"] // 2
        key = next((k for k in prof if k[0] in ("conv3d_k3_c48", "conv3d_k3_halo") and k[1:] == (Bg, R, E2, E2)), None) or ("conv3d_k3", Bg, R, E2, E2)
        evs = prof.get(key, [])
        kname = {"conv3d_k3_c48": "conv48_kernel (LDS-halo implicit GEMM", "conv3d_k3_halo": "conv_halo_kernel (LDS-halo implicit GEMM"}.get(
            key[0], "gemm_nt_kernel<bf16,4,3,AConv3> (generic gather implicit GEMM")
        if evs:
            ms = [a.elapsed_time(b) for a, b in evs]
            avg_eager = sum(ms) / len(ms)
            avg = avg_eager
            # The event pairs above sit in EAGER steps (graph replays cannot carry per-launch events), whose Python launch gaps let the part's
            # power management drop between kernels: on some boxes the 48->48 launches then average 4.1-4.2 ms while rocprofv3 of the
            # replayed step -- the timed region itself -- shows 3.6 ms (profiles/r2m_*: 0.39 vs 0.45 of peak in the same run).  `frac` stays
            # the eager-step figure (conservative); next to it the line carries the same kernel on the same shape launched ten times back
            # to back (`*_back_to_back`: no launch gaps, but also no neighbours sharing the power budget -- an upper bracket; the rocprofv3
            # per-launch average of the replayed step lies between the two).
            sustained = None
            try:
                if key[0] == "conv3d_k3_c48" and "decoder1.c1.wk" in model._pk.views:
                    xk = torch.randn((Bg, R, R, R, E2), device=dev).to(torch.bfloat16)
                    yk = torch.empty_like(xk)
                    wk = model._pk["decoder1.c1.wk"]
                    for _ in range(3):
                        ops.conv3d_k3_c48(xk, wk, out=yk)
                    torch.cuda.synchronize()
                    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ea.record()
                    for _ in range(10):
                        ops.conv3d_k3_c48(xk, wk, out=yk)
                    eb.record()
                    torch.cuda.synchronize()
                    sustained = ea.elapsed_time(eb) / 10
                    del xk, yk
            except Exception:  # noqa: BLE001
                sustained = None
            fl = 2.0 * 27 * E2 * E2 * (R ** 3) * Bg
            ach = fl / (avg * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "%s, conv3d 3x3x3 %d->%d @%d^3, fwd+dgrad launches)" % (kname, E2, E2, R),
                               "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                               "avg_launch_ms": round(avg, 4), "launches_timed": len(ms), "timing": "HIP event pairs around every launch of the kernel in %d eager steps" % ksteps,
                               "avg_launch_ms_back_to_back": None if sustained is None else round(sustained, 4),
                               "frac_back_to_back": None if sustained is None else round(fl / (sustained * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                               "traffic": None,
                               "algorithmic_flop_per_launch": fl, "algorithmic_bytes_per_launch": 2.0 * Bg * R ** 3 * E2 * 2}
            # the contract figure: the same kernel INSIDE the replayed graph (a second short run under rocprofv3 --kernel-trace); the eager-step
            # event figure stays in the line as `frac_eager_events`
            trace = replay_trace_kernels(args, Bg) if world == 1 else None
            if trace is not None:
                # the FAMILY average, launch-weighted: both instantiations of the 160^3 kernel (plain, and the input-gradient launch with the
                # InstanceNorm-backward sums in its epilogue) do the same 2*27*C*C FLOPs per voxel; the plain one alone stays in the line as frac_plain
                rx = "conv48_kernel<0, false, " if key[0] == "conv3d_k3_c48" else ("conv64_kernel" if key[0] == "conv3d_k3_halo" else None)
                hits = [(ns, cnt) for (nm, gx, gy, gz), (ns, cnt) in trace["kernels"].items() if rx and rx in nm and ns / max(cnt, 1e-9) > 1.0e6]
                plain = [(ns, cnt) for (nm, gx, gy, gz), (ns, cnt) in trace["kernels"].items() if "conv48_kernel<0, false, false" in nm and ns / max(cnt, 1e-9) > 1.0e6]
                if hits:
                    ns_tot, cnt_tot = sum(h[0] for h in hits), sum(h[1] for h in hits)
                    avg_r = ns_tot / cnt_tot / 1e6
                    rl = out["roofline"]
                    rl["frac_eager_events"], rl["avg_launch_ms_eager_events"] = rl["frac"], rl["avg_launch_ms"]
                    rl["avg_launch_ms"] = round(avg_r, 4)
                    rl["achieved"] = round(fl / (avg_r * 1e-3) / 1e12, 2)
                    rl["frac"] = round(fl / (avg_r * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)
                    rl["launches_timed"] = round(cnt_tot * trace["steps"])
                    if plain:
                        avg_p = sum(h[0] for h in plain) / sum(h[1] for h in plain) / 1e6
                        rl["frac_plain"], rl["avg_launch_ms_plain"] = round(fl / (avg_p * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), round(avg_p, 4)
                    rl["kernel"] = rl["kernel"].replace("fwd+dgrad launches)", "family average over the forward and the input-gradient (+ IN-backward sums) launches)")
                    rl["timing"] = ("rocprofv3 --kernel-trace of %d replayed steps of the SAME command at %d grids/GPU (a second short run spawned by bench.py; "
                                    "per-launch average of this kernel inside the HIP graph)" % (trace["steps"], Bg))
                try:
                    out["roofline"]["kernels"] = roofline_families(trace, cfg, R, Bg)
                except AssertionError as e:   # a mis-attributed family must not cost the line: it is reported instead of the table
                    out["roofline"]["kernels_error"] = str(e)
                out["roofline"]["replayed_step_ms_in_trace"] = round(trace["step_ns"] / 1e6, 3)
            # HBM traffic of this kernel comes from separate rocprofv3 --pmc passes (counters cannot be read in-process): the committed
            # summary is quoted only when it was taken on THIS kernel source (sha256 of conv48.hip) at the same shape
            try:
                tr = pmc_family_traffic("conv48", ("conv48_kernel<0, false, ",), Bg, R) if key[0] == "conv3d_k3_c48" else None
                if tr is not None:
                    out["roofline"].update(tr)
            except Exception as e:  # noqa: BLE001
                out["roofline"]["traffic_error"] = repr(e)
        # HBM-bound kernels: achieved GB/s of algorithmic bytes (ops.PROFILE_BYTES) against the 8 TB/s peak
        hb = []
        for k, evs2 in prof.items():
            nb = ops.PROFILE_BYTES.get(k)
            if nb:
                ms2 = [a.elapsed_time(b) for a, b in evs2]
                avg2 = sum(ms2) / len(ms2)
                hb.append({"kernel": k[0] + ":" + "x".join(str(v) for v in k[1:]), "ms_per_step": round(sum(ms2) / ksteps, 3), "avg_launch_ms": round(avg2, 4),
                           "algorithmic_bytes_per_launch": nb, "achieved_GBs": round(nb / (avg2 * 1e-3) / 1e9, 1),
                           "frac_of_8TBs": round(nb / (avg2 * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)})
        hb.sort(key=lambda d: -d["ms_per_step"])
        # rocprofv3-reported HBM bytes of the same kernels (separate --pmc passes, tools/pmc_step.sh), quoted only when they were taken
        # on this source of csrc/norm.hip at the same per-GPU batch
        try:
            kmap = {"mae_tail_fwd": ("tail_fwd",), "mae_tail_bwd": ("tail_bwd_kernel",), "instnorm_apply": ("in_apply_kernel",), "instnorm_bwd_apply": ("in_bwd_apply_kernel",),
                    "instnorm_bwd_apply_bg": ("in_bwd_apply_bg_kernel",),
                    "instnorm_bwd_reduce": ("in_reduce_kernel",)}
            for h in hb:
                pref = kmap.get(h["kernel"].split(":")[0])
                if not pref:
                    continue
                for fam in ("norm_streaming", "misc_streaming", "hbm_kernels"):
                    tr = pmc_family_traffic(fam, pref, Bg, R, pick="max")
                    if tr is not None and tr.get("traffic"):
                        h["pmc_hbm_bytes_per_launch"] = tr["traffic"]
                        h["pmc_GBs"] = round(tr["traffic"] / (h["avg_launch_ms"] * 1e-3) / 1e9, 1)
                        h["pmc_source"] = tr["traffic_source"]
                        break
        except Exception as e:  # noqa: BLE001
            out.setdefault("roofline", {})["hbm_pmc_error"] = repr(e)
        if "roofline" in out:
            out["roofline"]["hbm_kernels"] = hb[:3]
        tot = {}
        for k, evs2 in prof.items():
            tot[k[0] + ":" + "x".join(str(v) for v in k[1:])] = round(sum(a.elapsed_time(b) for a, b in evs2) / ksteps, 3)
        out["config"]["timed_kernel_ms_per_step"] = dict(sorted(tot.items(), key=lambda kv: -kv[1])[:8])

    if args.e2e and rank == 0 and world == 1:
        out["config"]["e2e"] = e2e_leg(model, args, R, sweep)

    if rank == 0 and world == 1 and args.dtype == "bf16" and not (args.no_sweep or args.eager) and os.environ.get("NMH_BENCH_INNER") != "1":
        fp = fp32_mode_leg(args)
        if fp:
            out["config"]["fp32_mode"] = fp
        if args.backbone == "swin_s" and os.environ.get("NMH_BENCH_OTHER_BACKBONES", "1") != "0":
            out["config"]["other_backbones"] = other_backbones_leg(args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, R, exts)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()



def replay_trace_kernels(args, Bg):
    """Per-kernel durations INSIDE the replayed HIP graph -- the timed region itself: a short second run of this script under
    `rocprofv3 --kernel-trace` (same model, batch and launch path; no sweep / CPU baseline / event timing), whose kernel trace is cut into
    steps at the weight-pack launches.  Returns {(kernel name, grid): [total ns per step, launches per step]} averaged over the replayed
    steps well inside the run, or None when rocprofv3 is not available (the line then keeps the eager-step event figures)."""
    import collections
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None or os.environ.get("NMH_BENCH_INNER") == "1" or os.environ.get("NMH_BENCH_NO_TRACE") == "1":
        return None
    out_dir = tempfile.mkdtemp(prefix="nmh_trace_", dir="/tmp")
    env = dict(os.environ, NMH_BENCH_INNER="1", TMPDIR="/tmp")
    cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", "bench", "--", sys.executable, os.path.abspath(__file__),
           "--batch-per-gpu", str(Bg), "--backbone", args.backbone, "--resolution", str(args.resolution), "--dtype", args.dtype,
           "--no-cpu-baseline", "--no-kernel-timing", "--no-sweep", "--steps", "8", "--warmup", "2"]
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600, check=True)
        f = glob.glob(out_dir + "/**/*kernel_trace.csv", recursive=True)[0]
        rows = list(csv.DictReader(open(f)))
    except Exception:  # noqa: BLE001
        shutil.rmtree(out_dir, ignore_errors=True)
        return None
    shutil.rmtree(out_dir, ignore_errors=True)
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "pack_kernel" in r["Kernel_Name"]]
    idx = [i for j, i in enumerate(idx) if j == 0 or int(rows[i]["Start_Timestamp"]) - int(rows[idx[j - 1]]["Start_Timestamp"]) > 5_000_000]
    if len(idx) < 7:
        return None
    steps = [(idx[k], idx[k + 1]) for k in range(len(idx) - 6, len(idx) - 2)]   # four replayed steps well inside the timed region
    agg = collections.defaultdict(lambda: [0.0, 0.0])
    for a, b in steps:
        for r in rows[a:b]:
            name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("void ", "").replace("unsigned short", "bf16")
            key = (name[:90], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
            agg[key][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / len(steps)
            agg[key][1] += 1.0 / len(steps)
    wall = sum(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]) for a, b in steps) / len(steps)
    return {"kernels": dict(agg), "step_ns": wall, "steps": len(steps)}


def roofline_families(trace, cfg, R, Bg):
    """the replayed step by kernel family: time per step from the kernel trace, algorithmic FLOPs per step from SURVEY 8(d)'s model, against the
    2.5 PFLOP/s dense bf16 MFMA peak (the HBM-bound families carry no FLOP figure)"""
    import re
    C, depths = cfg["embed_dim"], cfg["depths"]
    E2 = C // 2
    g = R // 4
    lin = attn = merge = 0.0
    s_ = g
    for i, d in enumerate(depths):
        c = C * 2 ** i
        if i > 0:
            s_ = (s_ + 1) // 2
            merge += 2.0 * s_ ** 3 * (4 * c) * c
        T = s_ ** 3
        lin += d * (2.0 * T * c * 3 * c + 2.0 * T * c * c + 16.0 * T * c * c)
        attn += d * 4.0 * T * 64 * c
    embed = 2.0 * g ** 3 * 256 * C
    conv_small = up = c3 = 0.0
    v = s_
    for cin, cout, k, skip in ((8 * C, 4 * C, 2, True), (4 * C, 2 * C, 2, True), (2 * C, C, 2, True), (C, C // 2, 4, False)):
        V = (v * k) ** 3
        up += 2.0 * V * cin * cout
        cc = 2 * cout if skip else cout
        if skip:
            conv_small += 2.0 * V * 27 * cc * cout + 2.0 * V * 27 * cout * cout
            c3 += 2.0 * V * cc * cout
        v *= k
    # blocks that ran the fused Swin-block forward kernels (csrc/swin_block.hip; dispatch rule of ops.swin_attn_ok / swin_mlp_ok at this batch)
    from nerf_mae_amd import ops as _ops
    sw_lin = sw_attn = 0.0
    s2 = g
    for i, d in enumerate(depths):
        c = C * 2 ** i
        if i > 0:
            s2 = (s2 + 1) // 2
        T = s2 ** 3
        nwin = Bg * ((s2 + 3) // 4) ** 3
        if _ops.SWIN_FUSED and c in _ops.SWIN_ATTN_WIDTHS and nwin >= _ops.SWIN_ATTN_MIN_WINDOWS:
            sw_lin += d * (2.0 * T * c * 3 * c + 2.0 * T * c * c)
            sw_attn += d * 4.0 * T * 64 * c
        if _ops.SWIN_FUSED and c in _ops.SWIN_MLP_WIDTHS and Bg * T >= _ops.SWIN_MLP_MIN_ROWS:
            sw_lin += d * 16.0 * T * c * c
    has_sw = any("swin_attn_fwd_kernel" in k[0] or "swin_mlp_fwd_kernel" in k[0] for k in trace["kernels"])
    if not has_sw:
        sw_lin = sw_attn = 0.0
    conv1 = 2.0 * 27 * E2 * E2 * R ** 3
    has_cc = any("cconv_fwd_kernel" in k[0] for k in trace["kernels"])
    has_cw = any("cconv_wgrad" in k[0] for k in trace["kernels"])   # (cconv_wgrad_kernel until round 4, cconv_wgrad_dma_kernel since)
    has_cd = any("cconv_dgrad_kernel" in k[0] for k in trace["kernels"])
    has_tc = any("tail_fwd_coarse_kernel" in k[0] for k in trace["kernels"])   # round 6: decoder1's transpose-conv forward runs inside the tail forward (no GEMM launch)
    up1_fwd = 2.0 * R ** 3 * C * (C // 2) if has_tc else 0.0
    composed = "(executes 216*96*48*2 FLOP per coarse cell -- a quarter of the reference's algorithmic FLOPs; frac_of_mfma_peak counts the EXECUTED ones)"
    fam = [
        ("decoder1 conv1 forward as ConvTranspose o conv composed on the coarse grid: cconv_fwd_kernel " + composed, r"cconv_fwd_kernel", conv1 if has_cc else None),
        ("decoder1 conv1 input gradient through the composition: cconv_dgrad_kernel " + composed, r"cconv_dgrad_kernel", conv1 if has_cd else None),
        ("decoder1 conv1 weight gradient through the composition: cconv_wgrad_kernel + reduce / border sums / chain rule to conv1 and the transpose conv "
         + composed, r"cconv_wgrad|cconv_dy_border", conv1 if has_cw else None),
        ("conv %d->%d 3x3x3 @%d^3 fwd+dgrad (conv48_kernel<0,false,*> / conv64_kernel; with the composed kernels: conv2 forward and conv2 input gradient, the latter "
         "with the InstanceNorm-backward sums in its epilogue, replacing a 1.2 ms reduce pass; round 6: conv2 reads z = lrelu(y1 - mean) with one weight image per sample -- the centered form)" % (E2, E2, R), r"conv48_kernel<0, false, (false|true)|conv64_kernel",
         (4 - int(has_cc) - int(has_cd)) * conv1),
        ("conv %d->%d 3x3x3 @%d^3 weight gradient (conv48_wgrad_kernel, persistent launches)" % (E2, E2, R), r"conv48_wgrad_kernel|conv64_wgrad_kernel",
         (1 if has_cw else 2) * conv1),
        ("decoder convs at the 10^3..40^3 levels, fwd+dgrad+wgrad (conv48_kernel<0,true>, AConv3, BConv3TN, small conv48_wgrad launches)",
         r"conv48_kernel<0, true, false|AConv3|BConv3TN|conv48_wgrad_reduce|conv48_wgrad_kernel", 3 * conv_small),   # (the 160^3 wgrad launch is taken by the family above)
        ("fused Swin-block forward kernels: LN1+QKV+window attention+proj+residual per window, LN2+fc1+GELU+fc2+residual per 64 tokens "
         "(swin_attn_fwd_kernel, swin_mlp_fwd_kernel) + their weight-stream pack", r"sw::swin_",
         (sw_lin + sw_attn) if has_sw else None),
        ("encoder Linear / patch-embed / merge / transpose-conv / 1x1 GEMMs fwd+dgrad (gemm_nt*, fused MLP)", r"gemm_nt|mlp96_|mlp_fwd_kernel|mlp_bwd_kernel|nt_ksplit|upconv4_fwd",
         2 * (lin + merge + up + c3) + embed - sw_lin - up1_fwd),
        ("encoder + transpose-conv weight gradients (gemm_tn_grouped, gemm_tn)", r"gemm_tn", lin + merge + embed + up + c3),
        ("window attention core fwd+bwd (attn_fwd / attn_bwd)", r"attn_", 3.5 * attn - sw_attn),
        ("LayerNorm fwd+bwd", r"ln_fwd|ln_bwd", None),
        ("decoder-1 elementwise passes @%d^3 (tail fwd/bwd, InstanceNorm backward; round 6: no stand-alone normalisation pass at %d^3, the tail forward forms the residual "
         "ConvT(x) itself on the matrix cores: 0.3 TFLOP not counted here) + the small levels' InstanceNorm launches" % (R, R),
         r"tail_fwd|tail_bwd|tail_sums|in_apply|in_bwd_apply|in_reduce|in_finalize|cconv_class_sums|cconv_mean|conv48_pack_scaled", None),
        ("weight pack (incl. the composed decoder1 weights) + grad norm + AdamW", r"pack_kernel|cconv_tr_kernel|cconv_dpack|upconv4_pack|tail_r_pack|adamw|sqnorm|clip_coef", None),
    ]
    used, out = set(), []
    ks = trace["kernels"]
    for name, rx, fl in fam:
        t = n = 0.0
        for key, (ns, cnt) in ks.items():
            if key in used or not re.search(rx, key[0]):
                continue
            big = ns / max(cnt, 1e-9) > 1.5e6      # the persistent 160^3 launches of the conv families
            if "weight gradient (conv48_wgrad_kernel" in name and not big:
                continue
            used.add(key)
            t += ns
            n += cnt
        row = {"family": name, "ms_per_step": round(t / 1e6, 3), "launches_per_step": round(n, 1)}
        if fl is not None and t > 0:
            fl_step = fl * Bg
            row["algorithmic_tflop_per_step"] = round(fl_step / 1e12, 4)
            if "cconv_" in rx:
                # the composed kernels EXECUTE a quarter of the reference's algorithmic FLOPs: hardware utilisation is the executed figure; the
                # algorithmic-equivalent rate (what the reference's two ops would need in this time) is kept under its own name
                row["executed_tflop_per_step"] = round(fl_step / 4 / 1e12, 4)
                row["achieved_tflops"] = round(fl_step / 4 / (t * 1e-9) / 1e12, 1)
                row["frac_of_mfma_peak"] = round(fl_step / 4 / (t * 1e-9) / 1e12 / PEAK_BF16_TFLOPS, 4)
                row["algorithmic_equiv_tflops"] = round(fl_step / (t * 1e-9) / 1e12, 1)
            else:
                row["achieved_tflops"] = round(fl_step / (t * 1e-9) / 1e12, 1)
                row["frac_of_mfma_peak"] = round(fl_step / (t * 1e-9) / 1e12 / PEAK_BF16_TFLOPS, 4)
        out.append(row)
    # sanity bound on the FLOP attribution: no family can exceed what an MFMA-only loop sustains on this part (0.70 of nominal); a
    # row above 0.75 means the family was credited FLOPs of launches it does not contain (round 5: a stale kernel-name match reported 0.81)
    bad = [r["family"][:60] for r in out if r.get("frac_of_mfma_peak", 0.0) > 0.75]
    assert not bad, "roofline_families: FLOP attribution above 0.75 of the MFMA peak for %s" % bad
    rest = sum(ns for key, (ns, cnt) in ks.items() if key not in used)
    out.append({"family": "everything else", "ms_per_step": round(rest / 1e6, 3), "launches_per_step": round(sum(c for k, (ns, c) in ks.items() if k not in used), 1)})
    return out

def other_backbones_leg(args):
    """BASELINE configs[1] and configs[3] on this GPU: swin_t and swin_b* (the defined deviation of SURVEY 8(c)) at 4 grids of 160^3 per step, bf16, each from a
    short child run of this script (3 warm-up + 5 timed replayed steps).  Informational rows of the N = 1 line; never `value`."""
    import subprocess
    env = dict(os.environ, NMH_BENCH_INNER="1")
    res = {}
    for bb in ("swin_t", "swin_b"):
        cmd = [sys.executable, os.path.abspath(__file__), "--batch-per-gpu", "4", "--backbone", bb, "--resolution", str(args.resolution), "--dtype", "bf16",
               "--no-cpu-baseline", "--no-kernel-timing", "--no-sweep", "--steps", "5", "--warmup", "3"]
        try:
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, check=True)
            d = json.loads(r.stdout.decode().strip().splitlines()[-1])
            res[bb] = {"grids_per_s": d["value"], "ms_per_step": d["ms_per_step"], "grids_per_gpu": 4, "dtype": "bf16", "steps": 5, "warmup": 3,
                       "algorithmic_tflop_per_grid_fwd_bwd": d["config"]["algorithmic_tflop_per_grid_fwd_bwd"],
                       "whole_step_mfma_frac": d["config"]["whole_step_mfma_frac"], "final_loss": d["config"]["final_loss"],
                       "baseline_config": "configs[1]" if bb == "swin_t" else "configs[3] (swin_b*: embed_dim 128, heads [4,8,16,32], SURVEY 8(c))"}
        except Exception as e:  # noqa: BLE001
            res[bb] = {"error": repr(e)[:200]}
    return res


def fp32_mode_leg(args):
    """Informational: the same step in the exact-fp32 mode (the reference's own arithmetic, SURVEY fact 4: every GEMM / conv on v_mfma_f32_16x16x4_f32,
    fp32 activations), 1 grid per step, from a short child run of this script.  Not the BASELINE metric (that is bf16) and never `value`."""
    import subprocess
    env = dict(os.environ, NMH_BENCH_INNER="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--batch-per-gpu", "1", "--backbone", args.backbone, "--resolution", str(args.resolution), "--dtype", "fp32",
           "--no-cpu-baseline", "--no-kernel-timing", "--no-sweep", "--steps", "6", "--warmup", "2"]
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, check=True)
        d = json.loads(r.stdout.decode().strip().splitlines()[-1])
        return {"grids_per_s": d["value"], "ms_per_step": d["ms_per_step"], "grids_per_gpu": 1, "dtype": "fp32",
                "note": "exact-fp32 mode of every kernel (parity mode; the reference trains in fp32), child run of this script; informational, not the BASELINE metric"}
    except Exception:  # noqa: BLE001
        return None


def e2e_leg(model, args, R, sweep):
    """Trainer.fit on HOST-resident synthetic scenes (stored format, through the pinned ring / copy stream / grid_prepare kernel):
    grids/s next to the HBM-resident figure of the same per-GPU batch"""
    import numpy as np
    from nerf_mae_amd import data
    from nerf_mae_amd.trainer import Trainer
    nb = 4
    res = {}
    for name, dt_ in (("uint8_scenes", np.uint8), ("fp32_scenes", np.float32)):
        scenes = [data.synthetic_scene((R, R, R), seed=50 + i, dtype=dt_) for i in range(8)]
        tr = Trainer(model, scenes * 32, batch_size=nb, num_epochs=1, log=lambda *_: None)   # 64 steps per epoch: the per-epoch pipeline fill
        # (first batch not overlapped, producer thread start-up, one loss read-back) is 20-40 ms, i.e. invisible in a real epoch and 2 ms/step in a 16-step one
        tr.train_epoch(1)      # capture + warm (pinned rings, copy stream)
        dt = None
        for ep in (2, 3):      # steady state: the better of two timed epochs of 64 steps
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr.train_epoch(ep)
            torch.cuda.synchronize()
            d1 = time.perf_counter() - t0
            dt = d1 if dt is None else min(dt, d1)
        gps = tr.steps_per_epoch * nb / dt
        res[name] = {"grids_per_s": round(gps, 2), "ms_per_step": round(1e3 * dt / tr.steps_per_epoch, 3), "grids_per_gpu": nb}
        ref = sweep.get("%d_grids_per_gpu" % nb)
        if ref:
            res[name]["frac_of_hbm_resident"] = round(gps / ref["grids_per_s"], 4)
        del tr
        torch.cuda.empty_cache()
    return res


def pmc_family_traffic(family, prefixes, Bg, R, pick="weighted"):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 --pmc summaries (profiles/*_<family>_pmc.json, written by tools/pmc_families.py /
    tools/pmc_kernel.sh on the GPU box: counters cannot be read in-process).  A summary is quoted only when the sha256 it carries matches the kernel source
    in this tree; bytes = 2 x FETCH_SIZE + WRITE_SIZE (the gfx950 correction of MI355X_MICROARCH.md), scaled from the grids per launch of the PMC run to Bg.
    Two formats: per-family files {source_file, source_sha256, grids_per_launch?, kernels{name: {hbm_bytes_per_launch, dispatches, mfma_busy_frac}}} (rounds 4+)
    and the flat round-2 files {conv48_hip_sha256 | norm_hip_sha256, batch_per_gpu, resolution, hbm_bytes_per_launch | kernels{}}.
    pick: "weighted" = dispatch-weighted mean over the matching kernels at full size, "max" = the largest (the 160^3 instantiation of a streaming kernel)."""
    import glob
    import re as _re
    stale = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*%s_pmc.json" % family)), key=os.path.basename)[::-1]:   # by name = round prefix, newest first (mtimes are arbitrary after a checkout)
        pm = json.load(open(f))
        srcf = pm.get("source_file")
        if srcf:
            want = pm.get("source_sha256")
        else:   # round-2 format
            srcf = "nerf-mae_amd/csrc/conv48.hip" if "conv48_hip_sha256" in pm else "nerf-mae_amd/csrc/norm.hip"
            want = pm.get("conv48_hip_sha256") or pm.get("norm_hip_sha256")
            if pm.get("resolution") not in (None, R):
                continue
        try:
            have = hashlib.sha256(open(os.path.join(ROOT, srcf), "rb").read()).hexdigest()
        except OSError:
            continue
        if have != want:
            stale.append(os.path.basename(f))
            continue
        g = pm.get("grids_per_launch") or pm.get("batch_per_gpu")
        if not g:
            m = _re.search(r"at (\d+) grids", pm.get("source", ""))
            g = int(m.group(1)) if m else None
        if not g:
            continue
        ks = pm.get("kernels") or {}
        hits = [(k, v) for k, v in ks.items() if any(k.startswith(p) for p in prefixes) and v.get("hbm_bytes_per_launch")]
        if not hits and "hbm_bytes_per_launch" in pm and family == "conv48":
            hits = [("conv48_kernel", pm)]
        if not hits:
            continue
        big = max(v["hbm_bytes_per_launch"] for _, v in hits)
        hits = [(k, v) for k, v in hits if v["hbm_bytes_per_launch"] > 0.25 * big]   # the full-size launches of the family (small decoder levels share the kernel)
        if pick == "max":
            per_launch, busy = big, None
        else:
            w = sum(v.get("dispatches", 1) for _, v in hits)
            per_launch = sum(v["hbm_bytes_per_launch"] * v.get("dispatches", 1) for _, v in hits) / w
            busy = [v["mfma_busy_frac"] * v.get("dispatches", 1) for _, v in hits if v.get("mfma_busy_frac")]
            busy = sum(busy) / w if busy else None
        res = {"traffic": per_launch / g * Bg,
               "traffic_source": "profiles/%s (2*FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes at %d grids per launch, scaled per grid; sha256 of %s matches this tree; kernels %s)" % (
                   os.path.basename(f), g, os.path.basename(srcf), [k for k, _ in hits][:3])}
        if busy:
            res["mfma_busy_frac_pmc"] = round(busy, 4)
        return res
    if stale:
        return {"traffic": None, "traffic_refused": "PMC summaries %s were taken on a different kernel source" % stale[:3]}
    return None


def cpu_baseline(args, R, exts):
    """CPU baseline: the oracle (a port of the reference's PyTorch path) on the host's physical cores,
    fp32, wall clock: swin_s 160^3 forward+backward, median of 3 repetitions after one warm-up pass at config 1's size; and config 1
    (swin_t, one 32^3 grid), median of 20 after 3 warm-ups."""
    from oracle import mae3d_oracle as O   # the only use of oracle/ in this file
    ncores = os.cpu_count() or 1
    try:
        import psutil
        ncores = psutil.cpu_count(logical=False) or ncores
    except Exception:  # noqa: BLE001
        pass
    torch.set_num_threads(ncores)

    def timed(ora, xg, reps, warm):
        ts = []
        for i in range(warm + reps):
            random.seed(i)
            for p in ora.parameters():
                p.grad = None
            tc = time.perf_counter()
            lo = ora(xg)
            lo[0].backward()
            if i >= warm:
                ts.append(time.perf_counter() - tc)
        return ts

    ora_t = O.build_oracle("swin_t", resolution=32, masking_prob=0.75, stochastic_depth_prob=0.1)
    ora_t.train()
    t_small = timed(ora_t, [O.synthetic_grid((32, 32, 32), seed=3)], 20, 3)
    ora = O.build_oracle(args.backbone, resolution=R, masking_prob=0.75, stochastic_depth_prob=0.1)
    ora.train()
    t_big = timed(ora, [O.synthetic_grid(exts[0], seed=7)], 3, 0)
    med = statistics.median(t_big)
    return {"value": round(1.0 / med, 5), "unit": "grids/s", "cores": ncores, "kind": "port",
            "sample": "1 grid %s %d^3 fp32 forward+backward, median of 3 repetitions (%s s), torch %s CPU" % (
                args.backbone, R, "/".join("%.1f" % t for t in t_big), torch.__version__),
            "config1_swin_t_32": {"value": round(1.0 / statistics.median(t_small), 3), "unit": "grids/s",
                                  "sample": "swin_t, one 32^3 grid, fp32 forward+backward, median of 20 repetitions after 3 warm-ups (%.3f s)" % statistics.median(t_small)}}


if __name__ == "__main__":
    sys.exit(main() or 0)
