import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch
from ramdsir import step as S
from oracle import step as OS, unet as OU
from golden_util import load_step, step_states
T = torch.from_numpy
name = 'fundus'
G, meta = load_step(os.path.join(ROOT, 'tests/golden'), name)
enc, dec, rec = step_states(meta)
bank, mods = S.make_bank('cuda:0', 3, 16, meta['num_classes'], len(meta['batch_sizes']))
for m, sd in (('enc', enc), ('dec', dec), ('rec', rec)):
    S.load_state(bank, m, sd)
ts = S.TrainStep(bank, mods, torch.float32, meta['batch_sizes'], meta['S'], meta['S'], dataset='fundus', consistency=meta['consistency'],
                 lr=meta['base_lr'], total_iters=meta['total_iters'], num_classes=meta['num_classes'])
ts.wpack.refresh()
ts.load_images(T(G['s0.img']).cuda(), T(G['s0.img_freq']).cuda()); ts.load_target(T(G['s0.mask']).cuda())
ts.step(); torch.cuda.synchronize()
# oracle with recorded BN outputs
rec_out = {}
orig_bn = OU._bn
def bn_hook(x, sd, name, training, domain=None):
    y = orig_bn(x, sd, name, training, domain)
    y.retain_grad()
    rec_out.setdefault((id(sd), name), []).append((domain, y))
    return y
OU._bn = bn_hook
cfg = OS.StepConfig(dataset='fundus', batch_sizes=meta['batch_sizes'], consistency=meta['consistency'], lr=meta['base_lr'],
                    total_iters=meta['total_iters'], num_classes=meta['num_classes'])
e2, d2, r2 = (OU.clone_state(s, requires_grad=True) for s in (enc, dec, rec))
loss, comps, inter = OS.forward_losses(e2, d2, r2, T(G['s0.img']), T(G['s0.img_freq']), T(G['s0.mask']), cfg)
loss.backward()
ids = {id(e2): 'enc', id(d2): 'dec', id(r2): 'rec'}
def cmp(plan):
    for node in reversed(plan.nodes):
        o = node.out
        if o.norm is None or o.g is None: continue
        key = [k for k in rec_out if ids[k[0]] == node.mname and k[1] == '.bn'.join(node.name.rsplit('.conv', 1))]
        lst = rec_out[key[0]]
        ref = torch.cat([y.grad for _, y in lst], 0)      # passes / domains in call order = image order
        got = o.g.float().cpu().permute(0, 3, 1, 2)
        fwd_ref = torch.cat([y.detach() for _, y in lst], 0)
        rms = float(ref.pow(2).mean().sqrt()) + 1e-30
        err = (got - ref).abs()
        per_img = err.flatten(1).max(1)[0] / rms
        print('%-4s %-14s g rel %.2e   per-image: %s' % (node.mname, node.name, float(err.max()) / rms, ' '.join('%.0e' % v for v in per_img.tolist())))
cmp(ts.seg) if len(sys.argv) < 2 else None
cmp(ts.rec)

def detail(plan, nm):
    for node in plan.nodes:
        if node.name != nm: continue
        o = node.out
        key = [k for k in rec_out if ids[k[0]] == node.mname and k[1] == '.bn'.join(node.name.rsplit('.conv', 1))]
        lst = rec_out[key[0]]
        ref = torch.cat([y.grad for _, y in lst], 0)
        got = o.g.float().cpu().permute(0, 3, 1, 2)
        rms = float(ref.pow(2).mean().sqrt())
        bad = ((got - ref).abs() > 1e-2 * rms).nonzero()
        print(node.mname, nm, 'bad elements', bad.shape[0], 'of', ref.numel())
        print(' images', sorted(set(bad[:, 0].tolist())), 'channels', sorted(set(bad[:, 1].tolist())))
        print(' rows', sorted(set(bad[:, 2].tolist())), 'cols', sorted(set(bad[:, 3].tolist())))
        for b in bad[:12].tolist():
            print('  ', b, 'got %.4e ref %.4e' % (got[tuple(b)].item(), ref[tuple(b)].item()))
detail(ts.seg, 'convu1.conv2')
detail(ts.rec, 'convu1.conv3')
