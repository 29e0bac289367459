"""Where do the HIP bf16 step and the oracle's bf16 rounding model (oracle.unet.rounding) part?  Layer by layer, one step on the
same seeded inputs: relative L2 of every raw conv output z (forward order) and of every gradient w.r.t. a BatchNorm output g
(backward order) -- HIP bf16 vs rounding-model oracle, next to (i) the rounding-model oracle against ITSELF with the input
images perturbed by one fp32 ulp (what any second correct implementation of the same rounding points looks like: a one-ulp
difference in an fp32 sum flips the bf16 rounding of a fraction of the stored values by a whole bf16 ulp, and every layer
re-quantises, so a perturbation eps grows like 0.04*sqrt(eps) per layer up to ~1e-3..1e-2), (ii) rounding-model oracle vs fp32
oracle (what the dtype itself does) and (iii) HIP fp32 vs fp32 oracle (the kernels' own noise).  A rounding point the model
misses would show as a JUMP of the first column at one layer kind, above the self column.
usage: bf16_gap.py [T128|C2|C3|C5]           (test infrastructure: imports oracle/ and tests/fullsize_util.py)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'ram-dsir_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch

import fullsize_util as FU
from oracle import step as OS, unet as OU

name = sys.argv[1] if len(sys.argv) > 1 else 'T128'
cfg = FU.CONFIGS[name]
src, trg, lam, mask = FU.synth(cfg)
states = FU.oracle_states(len(cfg['bs']))
img, frq = FU.oracle_ram(cfg, src, trg, lam)


def oracle_run(rounded, perturb=False):
    """-> {(module, conv name): [z per call]}, {(module, bn name): [g per call]} in call order."""
    im, fr = img, frq
    if perturb:                                             # one fp32 ulp up on every pixel: the smallest possible other input
        im, fr = torch.nextafter(img, torch.full_like(img, 2.0)), torch.nextafter(frq, torch.full_like(frq, 2.0))
    zs, gs = {}, {}
    ids = {}
    orig_conv, orig_bn = OU._conv, OU._bn

    def conv_hook(x, sd, nm, pad):
        z = orig_conv(x, sd, nm, pad)
        zs.setdefault((ids[id(sd)], nm), []).append(z)
        return z

    def bn_hook(x, sd, nm, training, domain=None, stats_from_stored=False):
        y = orig_bn(x, sd, nm, training, domain, stats_from_stored)
        if y.requires_grad:
            y.retain_grad()
        gs.setdefault((ids[id(sd)], nm), []).append(y)
        return y
    OU._conv, OU._bn = conv_hook, bn_hook
    try:
        c = OS.StepConfig(dataset=cfg['dataset'], batch_sizes=cfg['bs'], consistency='kd')
        e2, d2, r2 = (OU.clone_state(s, requires_grad=True) for s in states)
        ids.update({id(e2): 'enc', id(d2): 'dec', id(r2): 'rec'})
        if rounded:
            with OU.rounding(torch.bfloat16):
                loss, comps, inter = OS.forward_losses(e2, d2, r2, im, fr, torch.from_numpy(mask), c)
                loss.backward()
        else:
            loss, comps, inter = OS.forward_losses(e2, d2, r2, im, fr, torch.from_numpy(mask), c)
            loss.backward()
    finally:
        OU._conv, OU._bn = orig_conv, orig_bn
    z = {k: torch.cat([t.detach() for t in v], 0) for k, v in zs.items()}
    g = {k: torch.cat([t.grad for t in v], 0) for k, v in gs.items() if all(t.grad is not None for t in v)}
    return z, g


def hip_run(dtype):
    ts, bank, got = FU.hip_step(cfg, states, src, trg, lam, mask, dtype)
    z, g = {}, {}
    for plan in (ts.seg, ts.rec):
        for node in plan.nodes:
            if not hasattr(node, 'taps'):
                continue
            o = node.out
            z[(node.mname, node.name)] = o.buf[..., :o.C].float().cpu().permute(0, 3, 1, 2)
            if o.norm is not None and o.g is not None:
                g[(node.mname, '.bn'.join(node.name.rsplit('.conv', 1)))] = o.g[..., :o.C].float().cpu().permute(0, 3, 1, 2)
    return z, g, got


rel = FU.rel_l2
z32, g32 = oracle_run(False)
zb, gb = oracle_run(True)
zp, gp = oracle_run(True, perturb=True)
hz32, hg32, got32 = hip_run(torch.float32)
hzb, hgb, gotb = hip_run(torch.bfloat16)


def up_of(t_lo):
    return torch.nn.functional.interpolate(t_lo, scale_factor=2, mode='bilinear', align_corners=False)


print('== %s  forward: raw conv outputs z (rel L2)      HIP-bf16 vs model | model vs model(+1 ulp) | model vs fp32 oracle | HIP-fp32 vs fp32 oracle' % name)
bf = lambda t: t.to(torch.bfloat16).float()                  # the model's z is recorded before its storage rounding
for k in hzb:
    if k not in zb or hzb[k].shape != zb[k].shape:
        continue
    a, b, c, d = hzb[k], bf(zb[k]), z32[k], hz32[k]
    c = None if c.shape != a.shape else c                    # 1x1 conv: the fp32 oracle runs it above the upsample
    print('  %-4s %-14s %9.2e | %9.2e | %9.2e | %9.2e' % (k[0], k[1], rel(a, b), rel(bf(zp[k]), b), rel(b, c) if c is not None else float('nan'),
                                                          rel(d, c) if c is not None else float('nan')))
print('== backward: gradients w.r.t. BatchNorm outputs g (rel L2), backward order')
for k in reversed(list(hgb)):
    if k not in gb:
        continue
    a, b = hgb[k], gb[k]
    if a.shape != b.shape:
        continue
    c, d = g32.get(k), hg32.get(k)
    ok = c is not None and c.shape == a.shape
    print('  %-4s %-14s %9.2e | %9.2e | %9.2e | %9.2e' % (k[0], k[1], rel(a, b), rel(gp[k], b), rel(b, c) if ok else float('nan'), rel(d, c) if ok else float('nan')))
with OU.rounding(torch.bfloat16):
    refb = FU.oracle_step(cfg, states, img, frq, mask)
    refp = FU.oracle_step(cfg, states, torch.nextafter(img, torch.full_like(img, 2.0)), torch.nextafter(frq, torch.full_like(frq, 2.0)), mask)
med = lambda rows: float(np.median([r[0] for r in rows]))
print('parameter gradients, median rel L2 over tensors: HIP bf16 vs model %.3f | model vs model(+1 ulp) %.3f'
      % (med(FU.grad_table(gotb['grads'], refb['grads'])), med(FU.grad_table(refp['grads'], refb['grads']))))
