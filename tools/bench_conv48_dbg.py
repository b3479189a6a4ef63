"""timing decomposition of conv48_kernel via the NMH_C48_DBG diagnostic variants (one process per variant)"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    from nerf_mae_amd import ops
    from tests.test_kernels_gpu import _pack_via_kernel
    B, R, dt = 4, 160, torch.bfloat16
    x = torch.randn(B, R, R, R, 48, device='cuda').to(dt)
    w = torch.randn(48, 48, 3, 3, 3) * (27 * 48) ** -0.5
    wk = _pack_via_kernel(w, 6, dt, 41 * 3 * 64 * 8)
    y = torch.empty_like(x)
    dbg = int(os.environ.get('NMH_C48_DBG', '0'))
    buf = torch.zeros(256 * 8 * 16, dtype=torch.float64, device='cuda') if dbg & 32 else None
    fn = lambda: ops.conv3d_k3_c48(x, wk, out=y, stats_acc=buf)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    print(f"DBG={os.environ.get('NMH_C48_DBG', '0'):>3}: {a.elapsed_time(b) / 10:.4f} ms")
    if buf is not None:
        c = buf.view(torch.int64).view(256, 8, 16).double()
        names = ["ck0 compute", "ck0 vmcnt", "ck1 compute", "ck1 vmcnt", "ck2-4 compute", "ck2-4 vmcnt", "epilogue", "halo sstore", "tile barrier",
                 "ck0 barrier", "ck1 barrier", "ck2-4 barrier", "TOTAL"]
        tiles = 125.0
        m = c.mean(dim=(0, 1)) / tiles
        for i, n in enumerate(names):
            print(f"   {n:>14}: {m[i].item():9.0f} cycles/tile   (min wave {c[:, :, i].min().item() / tiles:9.0f}, max wave {c[:, :, i].max().item() / tiles:9.0f})")
        print("   per-wave totals of block 0:", (c[0, :, 12] / tiles).tolist())
else:
    for d in (0, 32):
        subprocess.run([sys.executable, __file__, "one"], env=dict(os.environ, NMH_C48_DBG=str(d)))
