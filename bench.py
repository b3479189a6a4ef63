#!/usr/bin/env python
"""bench.py -- NeRF-MAE hot-path benchmark on MI355X (contract: see the task brief / DESIGN.md "Measurement").

One "step" = zero_grad + forward + backward (+ gradient all-reduce when N>1) + clip + AdamW on one batch of synthetic
160^3 RGB-sigma grids already resident in HBM.  Metric: voxel-grids/s (whole job), swin_s, 160^3, bf16 (BASELINE.json).
Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: the 3x3x3 48->48 decoder conv, MFMA-bound) and
`cpu_baseline` (the CPU oracle timed on the host cores on a bounded sample).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import random
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--backbone", default="swin_s")
    ap.add_argument("--resolution", type=int, default=160)
    ap.add_argument("--batch-per-gpu", type=int, default=4,
                    help="grids per GPU per step (weak scaling).  Default 4 = the reference's training recipe (train_mae3d.sh: batch 32 on 8 GPUs; "
                         "BASELINE configs[1] is batch 4 on one GPU); 1 = BASELINE configs[2] read literally (global batch 8 at DP=8)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python each step instead of replaying the captured HIP graph")
    args = ap.parse_args()

    from nerf_mae_amd import ops
    from nerf_mae_amd.dist import GradReducer, broadcast_parameters
    from nerf_mae_amd.model import SWIN_CONFIGS, build_model
    from nerf_mae_amd.trainer import FusedAdamW, GraphedTrainStep, OneCycle
    from nerf_mae_amd import data

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    # dry-run hooks for a 1-GPU box (control flow of the N > 1 path: rank > 0 capture, flat all-reduce, optimizer graph): every rank
    # on device 0 and gloo instead of RCCL (which refuses two ranks on one GPU).  Never set by the driver.
    backend = os.environ.get("NMH_BENCH_BACKEND", "nccl")
    if os.environ.get("NMH_BENCH_SHARE_GPU", "0") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    R, Bg = args.resolution, args.batch_per_gpu
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    random.seed(0)
    model = build_model(args.backbone, resolution=R, masking_prob=0.75, stochastic_depth_prob=0.1, compute_dtype=dtype).to(dev)
    model.train()
    model.flatten_parameters()
    broadcast_parameters(model)
    reducer = GradReducer(model) if world > 1 else None
    model._reducer = reducer
    opt = FusedAdamW(model, lr=1e-4, weight_decay=1e-3, max_grad_norm=0.1)
    total_steps = args.steps + args.warmup
    sched = OneCycle(1e-4, max(total_steps, 2))

    # synthetic inputs: valid extents cycle through {160^3, 160x132x96, 120x160x144}, resident in HBM
    exts = [(R, R, R), (R, int(R * 0.825), int(R * 0.6)), (int(R * 0.75), R, int(R * 0.9))]
    # stored-format scenes (W,L,H,4 with raw density) go through the product input pipeline (density->alpha, layout, padding on the GPU)
    scenes = [data.synthetic_scene(exts[(rank * Bg + i) % 3], seed=rank * 131 + i) for i in range(Bg)]
    xb0, ext0 = data.GridBatcher(R, dev, normalize_density=True)(scenes, flags=[0] * Bg)
    grids = [xb0[i, :, :e[0], :e[1], :e[2]].contiguous() for i, e in enumerate(ext0.tolist())]
    mask_rng = random.Random(1000 + rank)
    g = R // 4

    from nerf_mae_amd.model import draw_block_mask
    graphed = None
    if not args.eager:
        model._reducer = None  # graph mode: one flat all-reduce between the backward graph and the optimizer graph
        graphed = GraphedTrainStep(model, opt, Bg, reducer=reducer)
        graphed(grids, draw_block_mask((g, g, g), 0.75, rng=mask_rng))  # loads the static batch, captures (lr is 0 until update_hyper)

    def step(i):
        lr, b1 = sched.at(i)
        opt.set_hyper(lr=lr, beta1=b1)
        bm = draw_block_mask((g, g, g), 0.75, rng=mask_rng)   # per-step python-random mask, as the reference
        if graphed is not None:
            return graphed(None, bm)[0]
        model.zero_grad()
        loss, l_rgb, l_a = model(grids, block_mask=bm)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step()
        return loss

    def barrier():
        if world > 1:
            if backend == "nccl":
                dist.barrier(device_ids=[local])
            else:
                torch.cuda.synchronize()
                dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        loss = step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    # per-kernel durations with HIP events on the launch stream: a few extra eager steps on the same model/data right after
    # the timed region (graph replays cannot carry per-launch events; the kernels, shapes and data are identical).  The side stream
    # is off for these steps so that an event pair brackets exactly one kernel (in the replayed step the decoder weight pack and the
    # weight-gradient GEMMs overlap other kernels; rocprofv3 of the replay shows the same per-launch time, profiles/)
    prof = None
    if not args.no_kernel_timing and rank == 0:
        side_was, ops.side_stream.enabled = ops.side_stream.enabled, False
        ops.PROFILE = {}
        ksteps = min(3, args.steps)
        for i in range(ksteps + 1):
            if i == 1:
                torch.cuda.synchronize()
                ops.PROFILE = {}       # first eager step after the replays: lazy allocations, not timed
            model.zero_grad()
            l3 = model(grids, block_mask=draw_block_mask((g, g, g), 0.75, rng=mask_rng))
            l3[0].backward()
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        ops.side_stream.enabled = side_was
    barrier()
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss_val = loss.item()

    grids_per_s = args.steps * Bg * world / dt
    cfg = SWIN_CONFIGS[args.backbone]
    flops_fb = 3.0 * fwd_flops_per_grid(cfg, R)

    out = {
        "metric": "voxel-grids/sec (fwd+bwd+optimizer) %s %d^3 %s" % (args.backbone, R, args.dtype),
        "value": round(grids_per_s, 4), "unit": "grids/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "%s MAE pretraining step, %d grid(s)/GPU of 4x%d^3 RGB-sigma, mask_ratio 0.75, stochastic depth 0.1, AdamW+clip, DP=%d"
                               % (args.backbone, Bg, R, world),
                   "global_batch": Bg * world, "resolution": R, "parallelism": "dp%d" % world, "launch": "eager" if args.eager else "hipgraph",
                   "algorithmic_tflop_per_grid_fwd_bwd": round(flops_fb / 1e12, 3),
                   "whole_step_mfma_frac": round(grids_per_s / world * flops_fb / (PEAK_BF16_TFLOPS * 1e12), 4), "final_loss": round(loss_val, 5)},
    }

    if rank == 0 and prof:
        # dominant kernel: implicit-GEMM 3x3x3 conv at R^3 with Cin=Cout=E/2 (decoder1 fwd + dgrad launches share one kernel)
        E2 = cfg["embed_dim"] // 2
        evs = prof.get(("conv3d_k3_c48", Bg, R, E2, E2), []) or prof.get(("conv3d_k3", Bg, R, E2, E2), [])
        kname = "conv48_kernel (LDS-halo implicit GEMM" if ("conv3d_k3_c48", Bg, R, E2, E2) in prof else "gemm_nt_kernel<bf16,4,3,AConv3> (generic gather implicit GEMM"
        if evs:
            ms = [a.elapsed_time(b) for a, b in evs]
            avg = sum(ms) / len(ms)
            fl = 2.0 * 27 * E2 * E2 * (R ** 3) * Bg
            ach = fl / (avg * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "%s, conv3d 3x3x3 %d->%d @%d^3, fwd+dgrad launches)" % (kname, E2, E2, R),
                               "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                               "avg_launch_ms": round(avg, 4), "launches_timed": len(ms), "traffic": None,
                               "algorithmic_flop_per_launch": fl}
            # HBM traffic of this kernel comes from separate rocprofv3 --pmc passes (counters cannot be read in-process); the
            # committed summary is used when it was taken at the same shape
            try:
                import glob
                for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*conv48_pmc.json")))[::-1]:
                    pm = json.load(open(f))
                    if pm.get("batch_per_gpu") == Bg and pm.get("resolution") == R and "c48" in kname.replace("conv48", "c48"):
                        out["roofline"]["traffic"] = pm["hbm_bytes_per_launch"]
                        out["roofline"]["traffic_source"] = os.path.basename(f) + " (2*FETCH_SIZE + WRITE_SIZE, separate PMC passes)"
                        out["roofline"]["algorithmic_bytes_per_launch"] = 2.0 * Bg * R ** 3 * E2 * 2
                        if pm.get("power"):   # measured context for `frac`: the kernel runs at the socket power cap
                            out["roofline"]["power_note"] = pm["power"]
                        break
            except Exception:
                pass
        tot = {}
        for k, evs in prof.items():
            tot[k[0] + ":" + "x".join(str(v) for v in k[1:])] = round(sum(a.elapsed_time(b) for a, b in evs) / ksteps, 3)
        out["config"]["timed_kernel_ms_per_step"] = dict(sorted(tot.items(), key=lambda kv: -kv[1])[:8])

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline: the oracle (a port of the reference's PyTorch path) on the host cores, bounded sample: ONE grid fwd+bwd
        from oracle import mae3d_oracle as O   # the only use of oracle/ in this file
        ncores = os.cpu_count() or 1
        try:
            import psutil
            ncores = psutil.cpu_count(logical=False) or ncores
        except Exception:
            pass
        torch.set_num_threads(ncores)
        ora = O.build_oracle(args.backbone, resolution=R, masking_prob=0.75, stochastic_depth_prob=0.1)
        ora.train()
        xg = [O.synthetic_grid(exts[0], seed=7)]
        random.seed(0)
        tc = time.perf_counter()
        lo = ora(xg)
        lo[0].backward()
        tcpu = time.perf_counter() - tc
        out["cpu_baseline"] = {"value": round(1.0 / tcpu, 5), "unit": "grids/s", "cores": ncores, "kind": "port",
                               "sample": "1 grid %s %d^3 fp32 forward+backward, 1 repetition (%.1f s), torch %s CPU" % (args.backbone, R, tcpu, torch.__version__)}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
