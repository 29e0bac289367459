import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch
from ramdsir import step as S
from oracle import step as OS, unet as OU
from golden_util import load_step, step_states
T = torch.from_numpy
name = sys.argv[1] if len(sys.argv) > 1 else 'fundus'
G, meta = load_step(os.path.join(ROOT, 'tests/golden'), name)
enc, dec, rec = step_states(meta)
bank, mods = S.make_bank('cuda:0', 3, 16, meta['num_classes'], len(meta['batch_sizes']))
for m, sd in (('enc', enc), ('dec', dec), ('rec', rec)):
    S.load_state(bank, m, sd)
ts = S.TrainStep(bank, mods, torch.float32, meta['batch_sizes'], meta['S'], meta['S'], dataset='fundus' if name.startswith('fundus') else 'prostate',
                 consistency=meta['consistency'], lr=meta['base_lr'], total_iters=meta['total_iters'], num_classes=meta['num_classes'])
ts.wpack.refresh()
ts.load_images(T(G['s0.img']).cuda(), T(G['s0.img_freq']).cuda()); ts.load_target(T(G['s0.mask']).cuda())
ts.step(); torch.cuda.synchronize()
cfg = OS.StepConfig(dataset='fundus' if name.startswith('fundus') else 'prostate', batch_sizes=meta['batch_sizes'], consistency=meta['consistency'],
                    lr=meta['base_lr'], total_iters=meta['total_iters'], num_classes=meta['num_classes'])
for dt in (torch.float32, torch.float64):
    e2, d2, r2 = (OU.clone_state(s, requires_grad=True) for s in (enc, dec, rec))
    if dt == torch.float64:
        for sd in (e2, d2, r2):
            for k in sd:
                if sd[k].is_floating_point():
                    sd[k] = sd[k].detach().double().requires_grad_(OU.is_param(k))
    mask = T(G['s0.mask'])
    mask = mask.to(dt) if mask.is_floating_point() else mask
    loss, comps, inter = OS.forward_losses(e2, d2, r2, T(G['s0.img']).to(dt), T(G['s0.img_freq']).to(dt), mask, cfg)
    loss.backward()
    print('==== oracle dtype', dt)
    for m, sd in (('enc', e2), ('dec', d2), ('rec', r2)):
        for k in OU.param_keys(sd):
            if k.endswith('.bias') and '.conv' in k: continue
            g = bank.g(m, k).cpu().double(); r = sd[k].grad.double()
            rms = float(r.pow(2).mean().sqrt()) + 1e-30
            print('%-4s %-28s rel %.2e   rms %.3e' % (m, k, float((g - r).abs().max()) / rms, rms))
