"""CPU-only: the noise floor of the REFERENCE arithmetic itself at a full-size configuration -- the fp32 oracle against (a) the
same run with inputs perturbed by ~1 ulp and (b) an fp64 run of the same code.  Output committed as profiles/r02_noise_floor_C2.txt;
it is what the gradient tolerances of tests/test_gpu_fullsize.py are derived from.  Test infrastructure (imports oracle/)."""
import sys, time, json
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch
import fullsize_util as FU
from oracle import step as OS, unet as OU
torch.set_num_threads(8)
name = sys.argv[1] if len(sys.argv) > 1 else 'C2'
cfg = FU.CONFIGS[name]
src, trg, lam, mask = FU.synth(cfg)
states = FU.oracle_states(len(cfg['bs']))
img, frq = FU.oracle_ram(cfg, src, trg, lam)
ref = FU.oracle_step(cfg, states, img, frq, mask)
print('fp32 oracle', ref['seconds'], flush=True)
# (a) 1-ulp-ish perturbation of the inputs (what a different but equally valid RAM rounding produces)
g = torch.Generator().manual_seed(0)
pert = lambda t: t + (torch.rand(t.shape, generator=g) - 0.5) * 2.4e-7
p = FU.oracle_step(cfg, states, pert(img), pert(frq), mask)
rows = FU.grad_table(p['grads'], ref['grads'])
rows.sort(reverse=True)
print('perturbed-input fp32 vs fp32: median %.3e worst %.3e %s' % (np.median([r[0] for r in rows]), rows[0][0], rows[0][2:]))
for r in rows[:6]: print('   %.3e %.3e %s.%s' % r)
print(' logit1 rel_l2 %.3e' % FU.rel_l2(p['logit1'], ref['logit1']))
# (b) fp64 run of the same code
st64 = tuple(type(s)((k, v.double() if v.is_floating_point() else v) for k, v in s.items()) for s in states)
c = OS.StepConfig(dataset=cfg['dataset'], batch_sizes=cfg['bs'], consistency='kd')
e2, d2, r2 = (OU.clone_state(s, requires_grad=True) for s in st64)
t0 = time.time()
m = torch.from_numpy(mask)
loss, comps, inter = OS.forward_losses(e2, d2, r2, img.double(), frq.double(), m.double() if m.is_floating_point() else m, c)
loss.backward()
print('fp64 oracle', time.time() - t0, flush=True)
g64 = {}
for mn, sd in (('enc', e2), ('dec', d2), ('rec', r2)):
    for k in OU.param_keys(sd):
        g64[(mn, k)] = sd[k].grad
rows = FU.grad_table(ref['grads'], g64)
rows.sort(reverse=True)
print('fp32 oracle vs fp64 oracle: median %.3e worst %.3e %s' % (np.median([r[0] for r in rows]), rows[0][0], rows[0][2:]))
for r in rows[:6]: print('   %.3e %.3e %s.%s' % r)
print(' logit1 rel_l2 %.3e' % FU.rel_l2(ref['logit1'], inter['logit1'].detach()))
