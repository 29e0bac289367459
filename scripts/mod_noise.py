"""How far do the drop-in modules' parameter gradients sit from the reference fixture, run to run?  (tests/test_gpu_modules.py)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch
import test_gpu_modules as TM
from golden_util import bn_shadowed_bias
M = np.load(os.path.join(ROOT, 'tests', 'golden', 'modules.npz'))
T = torch.from_numpy
res = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    enc, dec, rec = TM._models()
    for m in (enc, dec, rec): m.train()
    x = T(M['x']).to('cuda:0')
    feats = enc(x); logits = dec(feats)
    r1 = rec(feats[-1][0:2], domain_label=1 * torch.ones(2, dtype=torch.long))
    r2 = rec(feats[-1][2:4], domain_label=2 * torch.ones(2, dtype=torch.long))
    loss = (logits * T(M['wl']).cuda()).sum() + (r1 * T(M['wr0']).cuda()).sum() + (r2 * T(M['wr1']).cuda()).sum()
    loss.backward()
    worst, worst_norm, num, den, wk = 0.0, 0.0, 0.0, 0.0, ''
    for nm, m in (('enc', enc), ('dec', dec), ('rec', rec)):
        for k, p in m.named_parameters():
            if bn_shadowed_bias(k) or p.grad is None: continue
            g = p.grad.cpu()
            ref = M['train.g%s.sig.%s' % (nm, k)]
            nr = abs(float(g.double().norm()) / max(np.sqrt(ref[2]), 1e-30) - 1)
            worst_norm = max(worst_norm, nr)
            fk = 'train.g%s.full.%s' % (nm, k)
            if fk in M.files and np.abs(M[fk]).max() > 0:
                r = T(M[fk]).double()
                d = (g.double() - r)
                rl = float(d.norm() / r.norm())
                if rl > worst: worst, wk = rl, nm + '.' + k
                num += float(d.pow(2).sum()); den += float(r.pow(2).sum())
    res.append((worst, worst_norm, (num / den) ** 0.5, wk))
    print('rep %2d  worst per-tensor rel_l2 %.3e (%s)  worst norm dev %.3e  aggregate rel_l2 %.3e' % (rep, worst, wk, worst_norm, (num / den) ** 0.5), flush=True)
a = np.array([r[:3] for r in res])
print('max over runs: per-tensor %.3e  norm %.3e  aggregate %.3e' % tuple(a.max(0)))
