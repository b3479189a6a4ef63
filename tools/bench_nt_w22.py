"""tile sweep + correctness of the 2 x 2-wave NT GEMM (csrc/gemm.hip: gemm_nt_w22_kernel, NMH_GEMM_W22=<MT NT ST>) against the default dispatch on the
encoder's Linear shapes; graph replay over rotating buffers.  usage: python tools/bench_nt_w22.py [cfg,cfg,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_mae_amd import ops

dt = torch.bfloat16
NB = 4
cfgs = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [442, 443, 432, 433, 242, 243, 244, 233, 234, 423, 424]
# (name, M, N, K, epilogue kind)
SHAPES = [
    ("s2 dh      ", 8000, 1536, 384, "act2"),
    ("s2 dx1n    ", 8000, 384, 1536, "plain"),
    ("s2 do      ", 8000, 384, 384, "scale"),
    ("s2 dxn     ", 8000, 384, 1152, "plain"),
    ("s2 fc1 fwd ", 8000, 1536, 384, "act1"),
    ("s2 fc2 fwd ", 8000, 384, 1536, "resid"),
    ("s2 qkv fwd ", 13824, 1152, 384, "bias"),
    ("s3 dh      ", 1000, 3072, 768, "act2"),
    ("s3 dx1n    ", 1000, 768, 3072, "plain"),
    ("s3 dxn     ", 1000, 768, 2304, "plain"),
    ("s1 dh      ", 64000, 768, 192, "act2"),
    ("s1 dx1n    ", 64000, 192, 768, "plain"),
    ("s1 dxn     ", 64000, 192, 576, "plain"),
    ("s1 do      ", 64000, 192, 192, "scale"),
    ("s0 dxn     ", 512000, 96, 288, "plain"),
    ("4grid dx1n ", 4000, 384, 1536, "plain"),
    ("1grid dx1n ", 1000, 384, 1536, "plain"),
]


def make(M, N, K, kind):
    A = [(torch.randn(M, K, device="cuda") * 0.5).to(dt) for _ in range(NB)]
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
    out = [torch.empty(M, N, dtype=dt, device="cuda") for _ in range(NB)]
    kw = {}
    if kind in ("act1", "bias", "resid"):
        kw["bias"] = torch.randn(N, device="cuda")
    if kind in ("act1", "act2"):
        kw["act"] = 1 if kind == "act1" else 2
        kw["C2"] = (torch.randn(M, N, device="cuda")).to(dt)
    if kind in ("resid",):
        kw["resid"] = torch.randn(M, N, device="cuda").to(dt)
    if kind in ("scale", "resid"):
        kw["rowscale"] = torch.rand(8, device="cuda") + 0.5
        kw["rows_per_scale"] = (M + 7) // 8
    return A, W, out, kw


def reference(A, W, kw):
    v = A.float() @ W.float().T
    if "bias" in kw:
        v = v + kw["bias"]
    if kw.get("act") == 1:
        v = torch.nn.functional.gelu(v)
    elif kw.get("act") == 2:
        h = kw["C2"].float()
        v = v * (0.5 * (1 + torch.erf(h * 0.7071067811865476)) + h * 0.3989422804014327 * torch.exp(-0.5 * h * h))
    if "rowscale" in kw:
        rows = torch.arange(A.shape[0], device="cuda") // kw["rows_per_scale"]
        v = v * kw["rowscale"][rows][:, None]
    if "resid" in kw:
        v = v + kw["resid"].float()
    return v


def bench(A, W, out, kw, n=5):
    def run():
        for i in range(NB):
            k2 = dict(kw)
            if k2.get("act") == 1:
                k2["C2"] = torch.empty_like(out[i]) if False else kw["C2"]
            ops.gemm_nt(A[i], W, out=out[i], **k2)
    run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run(); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(n): run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (n * NB))
    return best * 1e3


for name, M, N, K, kind in SHAPES:
    A, W, out, kw = make(M, N, K, kind)
    if M * N <= 64000 * 768:
        ref = reference(A[0], W, kw)
    else:
        ref = None
    line = f"{name} M={M:6d} N={N:4d} K={K:4d} {kind:5s}:"
    for cfg in [None] + cfgs:
        os.environ["NMH_GEMM_W22"] = "0" if cfg is None else str(cfg)
        if kind == "act1":
            kw["C2"] = torch.empty(M, N, dtype=dt, device="cuda")
        try:
            us = bench(A, W, out, kw)
        except Exception as e:  # noqa
            line += f"  {cfg}: ERR({str(e)[:30]})"
            continue
        err = ""
        if ref is not None:
            ops.gemm_nt(A[0], W, out=out[0], **kw); torch.cuda.synchronize()
            d = (out[0].float() - ref).abs().max().item() / (ref.abs().max().item() + 1e-9)
            if d > 1e-2:
                err = f"!ERR {d:.1e}"
        fl = 2.0 * M * N * K / us / 1e6
        line += f"  {'def' if cfg is None else cfg}: {us:6.1f}us {fl:5.0f}TF{err}"
    print(line, flush=True)
    del A, W, out, kw, ref
    torch.cuda.empty_cache()
