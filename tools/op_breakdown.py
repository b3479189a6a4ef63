"""op-level (C-ABI call) GPU time of one eager training step, keyed by entry point + integer arguments"""
import sys, random, torch, collections
sys.path.insert(0, '.')
from nerf_mae_amd import ops
from nerf_mae_amd._lib import lib
from nerf_mae_amd.model import build_model, draw_block_mask
from oracle import mae3d_oracle as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ops.side_stream.enabled = False
m = build_model('swin_s', 160, 0.75, 0.1).cuda(); m.train(); m.flatten_parameters()
grids = [O.synthetic_grid((160, 160, 160), i).cuda() for i in range(B)]
bm = draw_block_mask((40, 40, 40), 0.75, rng=random.Random(0))
for it in range(3):
    if it == 2: lib().profile = {}
    m.zero_grad(); l = m(grids, block_mask=bm); l[0].backward(); torch.cuda.synchronize()
prof = lib().profile; lib().profile = None
rows = [(sum(a.elapsed_time(b) for a, b in v), len(v), k) for k, v in prof.items()]
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"B={B}: total op time {tot:.2f} ms over {sum(r[1] for r in rows)} calls")
byname = collections.defaultdict(lambda: [0.0, 0])
for t, n, k in rows: byname[k[0]][0] += t; byname[k[0]][1] += n
for k, (t, n) in sorted(byname.items(), key=lambda kv: -kv[1][0]): print(f"  {k:34s} {t:7.2f} ms {n:5d} calls")
print("top shapes:")
for t, n, k in rows[:70]: print(f"  {t:7.3f} ms {n:3d}x {k[0][4:]:28s} {k[1:]}")
