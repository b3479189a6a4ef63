import copy, random, sys, torch
sys.path.insert(0, '.')
from tests.test_model_gpu import _pair, _run_both, TINY, SWIN_T, relerr
from oracle import mae3d_oracle as O
for cfg, name, dt, res in [(SWIN_T, 'swin_t', torch.float32, 32)]:
    ora, hip = _pair(cfg, dt, res=res, init='default')
    xs = [O.synthetic_grid((res, res, res), 11), O.synthetic_grid((res - 2, res - 4, res), 12)]
    lo, lh = _run_both(ora, hip, xs, 42)
    o64 = copy.deepcopy(ora).double(); o64.zero_grad()
    g = res // 4
    bm = O.draw_block_mask((g, g, g), ora.masking_prob, rng=random.Random(42))
    l64 = o64([t.double() for t in xs], block_mask=bm.double(), return_pred=True); l64[0].backward()
    print(name, dt, 'loss64', l64[0].item(), 'ora32', lo[0].item(), 'hip', lh[0].item(), 'pred err hip/ora32 vs 64: %.2e %.2e' % (relerr(lh[3], l64[3]), relerr(lo[3], l64[3])))
    p64, po, ph = dict(o64.named_parameters()), dict(ora.named_parameters()), dict(hip.named_parameters())
    rows = []
    for n, p in p64.items():
        if p.grad is None or n.endswith(("conv1.bias", "conv2.bias", "conv3.bias")): continue
        a, b = ph[n].grad.float().cpu().flatten(), p.grad.float().flatten()
        rows.append((relerr(a, b), relerr(po[n].grad, p.grad), n, b.abs().max().item(), (torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)).item()))
    rows.sort(reverse=True)
    for r in rows[:10]: print('   hip %.3e ora32 %.3e %-50s max|g|=%.2e cos=%.6f' % r)
