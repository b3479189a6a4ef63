"""per-module-group GPU time (forward / backward) of one eager training step, measured with HIP events"""
import sys, random, torch
sys.path.insert(0, '.')
from nerf_mae_amd.model import build_model, draw_block_mask
from oracle import mae3d_oracle as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m = build_model('swin_s', 160, 0.75, 0.1).cuda(); m.train(); m.flatten_parameters()
grids = [O.synthetic_grid((160, 160, 160), i).cuda() for i in range(B)]
groups = {'embed': m.patch_partition, 'stage0': m.stages[0], 'stage1': m.stages[1], 'stage2': m.stages[2], 'stage3': m.stages[3],
          'dec4': m.decoder4, 'dec3': m.decoder3, 'dec2': m.decoder2, 'dec1': m.decoder1}
ev = {}
def rec(key):
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.setdefault(key, []).append(e)
# forward timing: wrap block-level modules
import types
def wrap(mod, name):
    mods = list(mod) if isinstance(mod, torch.nn.Sequential) and name.startswith('stage') else [mod]
    for sub in mods:
        of = sub.forward
        def f(*a, _of=of, **k):
            rec((name, 'f0')); r = _of(*a, **k); rec((name, 'f1')); return r
        sub.forward = f
        sub.register_full_backward_pre_hook(lambda mod_, go, n=name: rec((n, 'b0')))
        sub.register_full_backward_hook(lambda mod_, gi, go, n=name: rec((n, 'b1')))
for n, g in groups.items():
    if n != 'embed': wrap(g, n)
bm = draw_block_mask((40, 40, 40), 0.75, rng=random.Random(0))
for it in range(3):
    ev.clear(); m.zero_grad()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t2 = torch.cuda.Event(enable_timing=True)
    t0.record(); l = m(grids, block_mask=bm); t1.record(); l[0].backward(); t2.record(); torch.cuda.synchronize()
print(f"B={B} eager: forward {t0.elapsed_time(t1):.2f} ms, backward {t1.elapsed_time(t2):.2f} ms")
for n in groups:
    if n == 'embed': continue
    f = sum(a.elapsed_time(b) for a, b in zip(ev[(n, 'f0')], ev[(n, 'f1')]))
    b = sum(a.elapsed_time(b) for a, b in zip(ev.get((n, 'b0'), []), ev.get((n, 'b1'), [])))
    print(f"  {n:7s} fwd {f:7.2f} ms   bwd {b:7.2f} ms   ({len(ev[(n,'f0')])} module calls)")
