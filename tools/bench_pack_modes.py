"""time of the weight pack split by layout mode (swin_s): which modes cost what"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_mae_amd import ops
from nerf_mae_amd.model import build_model, _Packer
m = build_model('swin_s', 160, 0.75, 0.1).cuda(); m.train(); m.flatten_parameters(); m._ensure_ready(torch.device('cuda'))
P = m._packer
names = {0: "cast", 1: "transpose", 2: "conv fwd", 3: "conv dgrad", 4: "convT fwd", 5: "convT dgrad", 6: "c48 fwd", 7: "c48 dgrad"}
def timeit(f, n=20):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
b2d = P.blk2desc.cpu()
modes = torch.tensor([it[2] for it in P.items])
for mode in sorted(set(modes.tolist())):
    sel = (modes[b2d.long()] == mode).nonzero().flatten().cuda()
    bd, bs = P.blk2desc[sel].contiguous(), P.blkstart[sel].contiguous()
    n = sum(it[4] for it in P.items if it[2] == mode)
    t = timeit(lambda: ops.pack_weights(P.dt, P.descs, bd, bs, bd.numel()))
    print(f"mode {mode} {names[mode]:>11}: {n / 1e6:7.2f} M elements  {t * 1e3:7.1f} us")
def whole():
    P.run(); P.join()
print(f"whole pack (main + side stream, joined): {timeit(whole) * 1e3:.1f} us")
n = P.blk2desc.numel()
print(f"whole pack, one stream: {timeit(lambda: ops.pack_weights(P.dt, P.descs, P.blk2desc, P.blkstart, n)) * 1e3:.1f} us;  encoder part {timeit(lambda: ops.pack_weights(P.dt, P.descs, P.blk2desc[:P.split], P.blkstart[:P.split], P.split)) * 1e3:.1f} us")
