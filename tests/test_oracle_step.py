"""oracle.step vs. fixtures of three consecutive reference training steps (torch Adam, poly LR)."""
import numpy as np
import pytest
import torch

from oracle import step as OS, unet as OU
from golden_util import (assert_sig_close, load_step, step_states, state_checksum, bn_shadowed_bias,
                         assert_noise_level)

T = torch.from_numpy


@pytest.mark.parametrize('name', ['fundus', 'fundus_mse', 'prostate', 'fundus64', 'prostate96'])
def test_three_steps(golden_dir, name):
    G, meta = load_step(golden_dir, name)
    enc, dec, rec = step_states(meta)
    for c, nm in zip(map(state_checksum, (enc, dec, rec)), ('enc', 'dec', 'rec')):
        np.testing.assert_allclose(c, G['chk.' + nm], rtol=1e-12)
    cfg = OS.StepConfig(dataset='fundus' if name.startswith('fundus') else 'prostate',
                        batch_sizes=meta['batch_sizes'], lambda_rec=meta['lambda_rec'],
                        consistency=meta['consistency'], lr=meta['base_lr'], total_iters=meta['total_iters'],
                        num_classes=meta['num_classes'])
    opt = dict(enc=OS.adam_state({k: enc[k] for k in OU.param_keys(enc)}),
               dec=OS.adam_state({k: dec[k] for k in OU.param_keys(dec)}),
               rec=OS.adam_state({k: rec[k] for k in OU.param_keys(rec)}))
    for it in range(meta['nsteps']):
        lr_used = G['s%d.lr_used' % it]
        np.testing.assert_allclose([OS.poly_lr(cfg, it) / 2, OS.poly_lr(cfg, it), OS.poly_lr(cfg, it)], lr_used, rtol=1e-12)
        comps, grads = OS.train_step(enc, dec, rec, opt, T(G['s%d.img' % it]), T(G['s%d.img_freq' % it]),
                                     T(G['s%d.mask' % it]), cfg, it)
        ref = G['s%d.losses' % it]
        got = [comps[k].item() for k in ('seg1', 'dice1', 'seg2', 'dice2', 'cons', 'total')]
        # Step 0 is tight.  Adam's first update is lr*g/(|g|+1e-8) ~ lr*sign(g): elements whose true
        # gradient is below fp32 reassociation noise get a platform-dependent sign, i.e. a 2*lr kick
        # on weights of std ~0.03, so steps >= 1 agree only to ~1e-2 (measured 5e-3 on the 3-sample
        # DSBN slice) -- the reference itself is not reproducible beyond that across BLAS builds.
        np.testing.assert_allclose(got, ref, rtol=2e-5 if it == 0 else 1e-2)
        # per-domain restoration losses: one to three images per DomainSpecificBatchNorm group, 2x2-pixel BatchNorm batches at
        # the bottleneck -- the most host-sensitive scalars of the step after the first Adam update.  Measured against the
        # fixtures (generated in the build container): <= 5e-3 there, 1.5e-2 on the GPU box's host CPU (another core
        # count / ISA, so other fp32 reduction orders inside torch's CPU convs); this test also runs there
        # (tests/test_gpu_oracle_pinned.py), step 0 staying at 2e-5 on both hosts
        np.testing.assert_allclose(comps['rec'].numpy(), G['s%d.rec_losses' % it], rtol=2e-5 if it == 0 else 3e-2)
        if it == 0:
            for (gname, k), g in grads.items():
                if bn_shadowed_bias(k):
                    assert_noise_level(g, G['s0.g%s.sig.%s' % (gname, k.replace('.bias', '.weight'))], k)
                    continue
                assert_sig_close(g, G['s0.g%s.sig.%s' % (gname, k)], 5e-3, name=gname + '.' + k)
        if it == 0:
            # post-step parameters: Adam's first step moves every weight by ~lr*sign(g); BN-shadowed
            # conv biases have noise-level g so their sign is not reproducible -> skipped.
            for nm, sd in (('enc', enc), ('dec', dec), ('rec', rec)):
                for k, v in sd.items():
                    if bn_shadowed_bias(k):
                        continue
                    assert_sig_close(v.float(), G['s0.post.%s.sig.%s' % (nm, k)], 2e-3, name=nm + '.' + k)
