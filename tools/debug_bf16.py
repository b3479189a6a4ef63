import random, sys, torch
sys.path.insert(0, '.')
from tests.test_model_gpu import _pair, TINY, relerr
from nerf_mae_amd import model as M
from oracle import mae3d_oracle as O
ora, h32 = _pair(TINY, torch.float32)
_, h16 = _pair(TINY, torch.bfloat16)
xs = [O.synthetic_grid((32, 32, 32), 11), O.synthetic_grid((30, 28, 32), 12)]
bm = O.draw_block_mask((8, 8, 8), 0.75, rng=random.Random(42))
outs = {}
for name, m in (('f32', h32), ('b16', h16)):
    dev = m.mask_token.device
    m._ensure_ready(dev)
    xb, ext = m.transform([t.cuda() for t in xs], dev)
    mask_dev = bm.to(torch.uint8).view(-1).cuda()
    with torch.no_grad():
        tok = M._EmbedFn.apply(m._anchor, m, xb, mask_dev).view(2, 8, 8, 8, -1)
        rec = [('tok', tok)]
        x = tok
        feats = []
        for s, st in enumerate(m.stages):
            for i, mod in enumerate(st):
                x = mod(x)
                rec.append((f's{s}.{i}', x))
            feats.append(x)
        d = m.decoder4(feats[3], feats[2]); rec.append(('dec4', d))
        d = m.decoder3(d, feats[1]); rec.append(('dec3', d))
        d = m.decoder2(d, feats[0]); rec.append(('dec2', d))
        d = m.decoder1(d); rec.append(('dec1', d))
    outs[name] = rec
for (n, a), (_, b) in zip(outs['f32'], outs['b16']):
    print('%-8s %s relerr %.3e' % (n, tuple(a.shape), relerr(b, a)))
