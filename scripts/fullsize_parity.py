#!/usr/bin/env python3
"""Diagnostic: full-size parity numbers of the HIP step against the oracle (fp32) and of bf16 against HIP-fp32.
    python scripts/fullsize_parity.py C2 C3 C5 [--traj]   -> prints tables, writes gpurun_out/fullsize_parity.json
Test infrastructure (imports oracle/ and tests/fullsize_util.py)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'ram-dsir_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

import fullsize_util as FU


def show(rows, title, n=8):
    rows = sorted(rows, reverse=True)
    print('  %s: worst %d of %d (rel_l2, ref_rms, tensor); median %.2e' % (title, n, len(rows), float(np.median([r[0] for r in rows]))))
    for r in rows[:n]:
        print('    %.3e  %.3e  %s.%s' % r)
    return dict(worst=rows[0][0], median=float(np.median([r[0] for r in rows])), worst_key='%s.%s' % rows[0][2:])


def run(name, out):
    cfg = FU.CONFIGS[name]
    src, trg, lam, mask = FU.synth(cfg)
    states = FU.oracle_states(len(cfg['bs']))
    B = sum(cfg['bs'])
    print('== %s %s' % (name, cfg), flush=True)
    img, frq = FU.oracle_ram(cfg, src, trg, lam)
    ref = FU.oracle_step(cfg, states, img, frq, mask)
    print('  oracle fwd+bwd %.1f s; losses %s rec %s' % (ref['seconds'], ref['losses'], ref['rec']), flush=True)
    ts, bank, got = FU.hip_step(cfg, states, src, trg, lam, mask, torch.float32)
    scale = 127.5 if cfg['dataset'] == 'fundus' else 1.0
    ram_err = float((got['x'][B:] - frq).abs().max()) * scale
    img_err = float((got['x'][:B] - img).abs().max()) * scale
    res = dict(ram_max_abs_on_input_scale=ram_err, img_max_abs=img_err)
    print('  RAM: max|img_freq diff| %.3e (input scale), max|img diff| %.3e' % (ram_err, img_err))
    lrel = [abs(a - b) / abs(b) for a, b in zip(got['losses'], ref['losses'])]
    rrel = [abs(a - b) / abs(b) for a, b in zip(got['rec'], ref['rec'])]
    print('  losses hip %s' % got['losses'])
    print('  loss rel err %s rec rel err %s' % (['%.2e' % v for v in lrel], ['%.2e' % v for v in rrel]))
    res['loss_rel'] = max(lrel)
    res['rec_rel'] = max(rrel)
    for k in ('logit1', 'logit2'):
        rms = float(ref[k].pow(2).mean().sqrt())
        res[k] = dict(rel_l2=FU.rel_l2(got[k], ref[k]), max_over_rms=float((got[k] - ref[k]).abs().max()) / rms)
        print('  %s: rel_l2 %.3e  max|d|/rms %.3e' % (k, res[k]['rel_l2'], res[k]['max_over_rms']))
    rs = torch.tanh(got['rec_logits'])
    res['rec_soft_max_abs'] = float((rs - ref['rec_soft']).abs().max())
    print('  rec_soft max|d| %.3e' % res['rec_soft_max_abs'])
    res['grads_fp32_vs_oracle'] = show(FU.grad_table(got['grads'], ref['grads']), 'fp32 HIP vs oracle grads')
    # network parity with RAM taken out: feed the oracle's images
    ts2, bank2, got2 = FU.hip_step(cfg, states, src, trg, lam, mask, torch.float32, given_images=(img, frq))
    res['grads_fp32_given_images'] = show(FU.grad_table(got2['grads'], ref['grads']), 'fp32 HIP (oracle images) vs oracle grads')
    res['grads_fp32_run_to_run'] = show(FU.grad_table(got2['grads'], got['grads']), 'fp32 HIP run (oracle images) vs run (GPU RAM)')
    del ts2, bank2
    # bf16 vs HIP fp32
    ts3, bank3, got3 = FU.hip_step(cfg, states, src, trg, lam, mask, torch.bfloat16)
    lrel = [abs(a - b) / abs(b) for a, b in zip(got3['losses'], got['losses'])]
    print('  bf16 losses %s\n  bf16 vs fp32 loss rel %s' % (got3['losses'], ['%.2e' % v for v in lrel]))
    res['bf16_loss_rel'] = max(lrel)
    for k in ('logit1', 'logit2'):
        res['bf16_' + k] = FU.rel_l2(got3[k], got[k])
        print('  bf16 %s rel_l2 vs fp32 %.3e' % (k, res['bf16_' + k]))
    res['grads_bf16_vs_fp32'] = show(FU.grad_table(got3['grads'], got['grads']), 'bf16 HIP vs fp32 HIP grads')
    # bf16 HIP vs the oracle with the SAME rounding points (oracle.unet.rounding)
    from oracle import unet as OU
    with OU.rounding(torch.bfloat16):
        refb = FU.oracle_step(cfg, states, img, frq, mask)
    lrel = [abs(a - b) / abs(b) for a, b in zip(got3['losses'], refb['losses'])]
    print('  bf16 HIP vs bf16-rounding oracle: loss rel %s' % ['%.2e' % v for v in lrel])
    res['bf16m_loss_rel'] = max(lrel)
    for k in ('logit1', 'logit2'):
        res['bf16m_' + k] = FU.rel_l2(got3[k], refb[k])
        print('  bf16 HIP %s rel_l2 vs bf16-rounding oracle %.3e   (that oracle vs fp32 oracle: %.3e)' % (k, res['bf16m_' + k], FU.rel_l2(refb[k], ref[k])))
    res['grads_bf16_vs_model'] = show(FU.grad_table(got3['grads'], refb['grads']), 'bf16 HIP vs bf16-rounding oracle grads')
    res['grads_model_vs_fp32'] = show(FU.grad_table(refb['grads'], ref['grads']), 'bf16-rounding oracle vs fp32 oracle grads')
    out[name] = res
    del ts, bank, ts3, bank3
    torch.cuda.empty_cache()


def trajectory(out, name='T128', nsteps=20):
    """Fixed batch, nsteps steps: bf16 HIP and fp32 HIP loss trajectories against the fp32 oracle (torch Adam math)."""
    from oracle import step as OS, unet as OU
    cfg = FU.CONFIGS[name]
    src, trg, lam, mask = FU.synth(cfg)
    states = FU.oracle_states(len(cfg['bs']))
    img, frq = FU.oracle_ram(cfg, src, trg, lam)
    enc, dec, rec = (OU.clone_state(s) for s in states)
    opt = dict(enc=OS.adam_state({k: enc[k] for k in OU.param_keys(enc)}), dec=OS.adam_state({k: dec[k] for k in OU.param_keys(dec)}),
               rec=OS.adam_state({k: rec[k] for k in OU.param_keys(rec)}))
    c = OS.StepConfig(dataset=cfg['dataset'], batch_sizes=cfg['bs'], consistency='kd', lr=2e-3, total_iters=1000)
    ref = []
    for it in range(nsteps):
        comps, _ = OS.train_step(enc, dec, rec, opt, img, frq, torch.from_numpy(mask), c, it)
        ref.append(comps['total'].item())
    hist = {}
    for dt, nm in ((torch.float32, 'f32'), (torch.bfloat16, 'bf16')):
        ts, bank, got = FU.hip_step(cfg, states, src, trg, lam, mask, dt, total_iters=1000, nsteps=nsteps)
        hist[nm] = [h[5] for h in got['hist']]
        del ts, bank
    print('== trajectory %s' % name)
    for it in range(nsteps):
        print('  it %2d  oracle %.5f  f32 %.5f (%.2e)  bf16 %.5f (%.2e)' % (it, ref[it], hist['f32'][it], abs(hist['f32'][it] / ref[it] - 1),
                                                                          hist['bf16'][it], abs(hist['bf16'][it] / ref[it] - 1)))
    out['traj_' + name] = dict(oracle=ref, f32=hist['f32'], bf16=hist['bf16'])


if __name__ == '__main__':
    names = [a for a in sys.argv[1:] if not a.startswith('--')] or ['C2']
    out = {}
    for n in names:
        run(n, out)
    if '--traj' in sys.argv:
        trajectory(out)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'fullsize_parity.json'), 'w') as f:
        json.dump(out, f, indent=1)
