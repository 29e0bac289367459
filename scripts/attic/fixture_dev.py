"""Per-tensor deviation of the HIP fp32 step from a reference step fixture (sorted).  usage: fixture_dev.py <name>"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'ram-dsir_amd')]
import numpy as np, torch
import test_gpu_step as TG
name = sys.argv[1] if len(sys.argv) > 1 else 'prostate96'
G, meta, states, bank, mods, ts = TG._setup(os.path.join(ROOT, 'tests', 'golden'), name, torch.float32)
TG._feed(ts, G, 0)
ts.step()
torch.cuda.synchronize()
print('losses', [ts.losses[i].item() for i in range(5)], 'ref', list(G['s0.losses'][:5]))
rows = []
for m in ('enc', 'dec', 'rec'):
    for key, shape, kind, _ in dict(mods)[m]:
        if kind != 'param' or TG.bn_shadowed_bias(key):
            continue
        fk = 's0.g%s.full.%s' % (m, key)
        g = bank.g(m, key).cpu()
        if fk in G.files:
            rows.append((TG.rel_l2(g, torch.from_numpy(G[fk])), m + '.' + key, 'full'))
        else:
            ref = G['s0.g%s.sig.%s' % (m, key)]
            rows.append((abs(float(g.double().norm()) / np.sqrt(ref[2]) - 1), m + '.' + key, 'norm'))
rows.sort(reverse=True)
for r in rows[:14]:
    print('%.3e  %s (%s)' % r)
print('median full %.2e, median norm %.2e' % (np.median([r[0] for r in rows if r[2] == 'full']), np.median([r[0] for r in rows if r[2] == 'norm'])))
