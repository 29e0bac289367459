"""-m gpu: ONE training step at the BASELINE.json configurations -- C2 (Fundus 400x400, batch 8 = [2,3,3]), C3 (Prostate 384x384,
batch 10 = [2]*5, CE / softmax / dice_loss_multi) and C5 (512x512, 4 domains) -- with RAM on the GPU, through the DEFAULT kernel
dispatch (persistent small-channel convs with multi-tile ranges, conv_pp, XCD block remap, 64/32-wide tile switch, split
weight gradients: none of which a 32x32 fixture reaches), against the oracle run live on the same seeded inputs
(reference: code/train.py:246-296 / :412-465, code/dataset/fundus.py:41-61).

Tolerances and where they come from
  fp32   RAM output 2e-3 on the 0..255 scale; the five loss terms and the per-domain rec losses 1e-4 relative; logits
         relative L2 1e-4 and max|d| <= 1e-3 RMS (measured 1.3e-5 / 1.9e-4).
         Parameter gradients: relative L2 per tensor <= 2e-2, median over tensors <= 6e-3 (measured: worst 9.5e-3, median
         3.3e-3).  That is the noise floor of the REFERENCE arithmetic itself at this size, not of the kernels:
         scripts/noise_floor.py (output committed: profiles/r02_noise_floor_C2.txt) runs the fp32 oracle against
         itself with inputs perturbed by one ulp -> median 2.9e-3, worst 8.1e-3; against an fp64 run of the same code ->
         median 4.5e-3, worst 9.3e-2.  A perturbation of relative size d flips a fraction ~d of the ReLU / max-pool
         decisions and each flip is an O(1) change of that element's gradient: relative L2 ~ sqrt(d) (d ~ 1e-5 after ~25
         BatchNorm'd layers -> 3e-3).  Conv biases in front of a train-mode BatchNorm are excluded (exact zeros here, fp32
         cancellation noise in the reference).
  bf16   (the dtype of the bench line) same sqrt law with d ~ bf16 ulp accumulated over the layers: element-wise agreement
         with fp32 is not defined (logits 12 %, gradients ~40 % relative L2 -- and the SAME numbers separate the fp32 oracle
         from the oracle run with bf16 rounding at the HIP path's rounding points, oracle.unet.rounding).  Asserted instead:
         losses within 2e-3 of that rounding-model oracle (measured 5e-4) and 3e-2 of fp32; logits closer to the rounding
         model than the rounding model is to fp32; the full gradient's direction (cosine >= 0.97 vs the model; measured 0.986-0.991) and
         its distance from the model's own one-ulp self-distance, at every configuration (C2 / C3 / C5 / F256);
         and a 20-step fixed-batch loss trajectory inside a band around the fp32 oracle's (torch-Adam restatement).
"""
import functools

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import fullsize_util as FU                                   # noqa: E402


@functools.lru_cache(maxsize=None)
def _case(name, rounded=False, perturbed=False):
    """Inputs + oracle results of a configuration (cached: the fp32 and bf16 tests of C2 share them).  perturbed: the network
    inputs moved up by ONE fp32 ulp -- the smallest possible 'other correct implementation' of the same arithmetic."""
    from oracle import unet as OU
    cfg = FU.CONFIGS[name]
    src, trg, lam, mask = FU.synth(cfg)
    states = FU.oracle_states(len(cfg['bs']))
    img, frq = FU.oracle_ram(cfg, src, trg, lam)
    if perturbed:
        img, frq = torch.nextafter(img, torch.full_like(img, 2.0)), torch.nextafter(frq, torch.full_like(frq, 2.0))
    if rounded:
        with OU.rounding(torch.bfloat16):
            ref = FU.oracle_step(cfg, states, img, frq, mask)
    else:
        ref = FU.oracle_step(cfg, states, img, frq, mask)
    return cfg, (src, trg, lam, mask), states, (img, frq), ref


def _flat(grads, keys):
    return torch.cat([grads[k].double().reshape(-1) for k in keys])


# F256 = the reference's own Fundus training shape: batch 16 = [3, 6, 7] at 256 x 256 (code/train.py:35,541): uneven DSBN groups
# of 3 / 6 / 7 images through the default dispatch -- what a user's first real run executes
@pytest.mark.parametrize('name', ['C2', 'C3', 'C5', 'F256'])
def test_fullsize_fp32_step_matches_oracle(name):
    cfg, (src, trg, lam, mask), states, (img, frq), ref = _case(name)
    ts, bank, got = FU.hip_step(cfg, states, src, trg, lam, mask, torch.float32)
    B = sum(cfg['bs'])
    scale = 127.5 if cfg['dataset'] == 'fundus' else 1.0            # back to the scale RAM works on (0..255 / [-1,1])
    assert float((got['x'][B:] - frq).abs().max()) * scale < 2e-3    # RAM (rd_ram_mix) vs fundus.py:41-61 in float64
    assert float((got['x'][:B] - img).abs().max()) * scale < 1e-4
    np.testing.assert_allclose(got['losses'], ref['losses'], rtol=1e-4)
    np.testing.assert_allclose(got['rec'], ref['rec'], rtol=1e-4)
    for k in ('logit1', 'logit2'):
        rms = float(ref[k].pow(2).mean().sqrt())
        assert FU.rel_l2(got[k], ref[k]) < 1e-4, k
        assert float((got[k] - ref[k]).abs().max()) < 1e-3 * rms, k
    assert float((torch.tanh(got['rec_logits']) - ref['rec_soft']).abs().max()) < 3e-3
    rows = FU.grad_table(got['grads'], ref['grads'])
    assert len(rows) >= 160
    bad = [r for r in rows if not r[0] <= 2e-2]
    assert not bad, sorted(bad, reverse=True)[:8]
    assert float(np.median([r[0] for r in rows])) <= 6e-3
    for (m, k), r in ref['grads'].items():                           # analytically zero (docstring)
        if FU.bn_shadowed_bias(k):
            assert float(got['grads'][(m, k)].abs().max()) == 0.0, (m, k)
    # the whole flat gradient
    keys = [(r[2], r[3]) for r in rows]
    assert FU.rel_l2(_flat(got['grads'], keys), _flat(ref['grads'], keys)) < 6e-3


@pytest.mark.parametrize('name', ['C2', 'C3', 'C5', 'F256'])
def test_fullsize_bf16_step_against_the_rounding_model_oracle(name):
    """bf16 -- the dtype of the bench line -- at every BASELINE.json shape: C2, C3 (softmax / CE / dice_loss_multi, five
    single-pair DSBN groups at 384x384) and C5 (512x512, four domains), and at the reference's native training shape F256."""
    cfg, (src, trg, lam, mask), states, (img, frq), ref32 = _case(name)
    _, _, _, _, refb = _case(name, rounded=True)
    ts, bank, got = FU.hip_step(cfg, states, src, trg, lam, mask, torch.bfloat16)
    np.testing.assert_allclose(got['losses'], refb['losses'], rtol=2e-3)
    np.testing.assert_allclose(got['rec'], refb['rec'], rtol=2e-3)
    np.testing.assert_allclose(got['losses'], ref32['losses'], rtol=3e-2)
    for k in ('logit1', 'logit2'):
        d_model, d_dtype = FU.rel_l2(got[k], refb[k]), FU.rel_l2(refb[k], ref32[k])
        assert d_model < 8e-2 and d_model < 0.7 * d_dtype, (k, d_model, d_dtype)
    rows = FU.grad_table(got['grads'], refb['grads'])
    rows_dtype = FU.grad_table(refb['grads'], ref32['grads'])
    med, med_dtype = float(np.median([r[0] for r in rows])), float(np.median([r[0] for r in rows_dtype]))
    keys = [(r[2], r[3]) for r in rows]
    a, b, c = _flat(got['grads'], keys), _flat(refb['grads'], keys), _flat(ref32['grads'], keys)
    cos = lambda u, v: float((u @ v) / (u.norm() * v.norm()))
    # The yardstick for "how close can two correct implementations of these rounding points be": the rounding-model oracle
    # against ITSELF on inputs one fp32 ulp away.  A one-ulp difference in an fp32 sum flips the bf16 rounding of a fraction
    # of the stored values by a whole bf16 ulp, and every layer re-quantises: a perturbation eps grows like 0.04 sqrt(eps)
    # per layer until it saturates near 1e-2 (scripts/bf16_gap.py, profiles/r03_bf16_gap_C2.txt: layer by layer, no jump at
    # any layer kind -- the HIP path misses no rounding point of the model).  The kernels must sit at that self-distance, at
    # EVERY configuration (a broken rounding point would pass the absolute gates below at C3 / C5 / F256 otherwise).
    _, _, _, _, refp = _case(name, rounded=True, perturbed=True)
    med_self = float(np.median([r[0] for r in FU.grad_table(refp['grads'], refb['grads'])]))
    bp = _flat(refp['grads'], keys)
    print('bf16 %s: median grad rel-L2 %.3f (dtype %.3f, self %.3f), cos(hip, model) %.4f cos(hip, fp32) %.4f cos(model, fp32) %.4f '
          'cos(self, model) %.4f' % (name, med, med_dtype, med_self, cos(a, b), cos(a, c), cos(b, c), cos(bp, b)))
    # gates at what is measured (round 4, MI355X; C2 / C3 / C5 / F256): median 0.217 / 0.254 / 0.187 / 0.200 (the model's own one-ulp
    # self-distance: 0.158 / 0.259 / 0.164 / 0.193), cos(hip, model) 0.986 / 0.988 / 0.991 / 0.986, cos(hip, fp32) 0.949 / 0.961 /
    # 0.969 / 0.950 (cos(model, fp32) the same to 0.002) -- round 3 asserted 0.35 / 0.85 / 0.80
    assert med < 0.8 * med_dtype and med < 0.30, (med, med_dtype)    # the kernels add less than the dtype itself does
    assert cos(a, b) >= 0.97 and cos(a, c) >= 0.93, (cos(a, b), cos(a, c), cos(b, c))
    assert abs(cos(a, c) - cos(b, c)) <= 0.01, (cos(a, c), cos(b, c))   # as far from fp32 as the rounding model is, not further
    assert abs(float(a.norm() / b.norm()) - 1.0) < 0.1
    assert med <= 1.6 * med_self + 0.02, (med, med_self)
    assert cos(a, b) >= cos(bp, b) - 0.05, (cos(a, b), cos(bp, b))


def test_bf16_and_fp32_loss_trajectories_track_the_oracle():
    """20 steps on one fixed batch at 128x128 ([2,3,3], RAM on, lr 2e-3 poly over 1000 iterations): total loss per step of
    the HIP path in fp32 and in bf16 against the oracle's (fp32, torch-Adam math, train.py:285-293).  Measured: both within
    a few % of the oracle for the first 8 steps and within 1e-1 afterwards, mean |log ratio| 0.03 (Adam's sign-like early updates make the
    trajectories diverge chaotically at the few-% level -- the fp32 HIP path and the oracle do so between themselves);
    the loss falls from 5.75 to 0.12."""
    from oracle import step as OS, unet as OU
    cfg = FU.CONFIGS['T128']
    src, trg, lam, mask = FU.synth(cfg)
    states = FU.oracle_states(len(cfg['bs']))
    img, frq = FU.oracle_ram(cfg, src, trg, lam)
    enc, dec, rec = (OU.clone_state(s) for s in states)
    opt = {m: OS.adam_state({k: sd[k] for k in OU.param_keys(sd)}) for m, sd in (('enc', enc), ('dec', dec), ('rec', rec))}
    c = OS.StepConfig(dataset='fundus', batch_sizes=cfg['bs'], consistency='kd', lr=2e-3, total_iters=1000)
    nsteps = 20
    ref = []
    for it in range(nsteps):
        comps, _ = OS.train_step(enc, dec, rec, opt, img, frq, torch.from_numpy(mask), c, it)
        ref.append(comps['total'].item())
    for dt in (torch.float32, torch.bfloat16):
        ts, bank, got = FU.hip_step(cfg, states, src, trg, lam, mask, dt, total_iters=1000, nsteps=nsteps)
        hist = [h[5] for h in got['hist']]
        assert all(np.isfinite(hist))
        # step 0 is the same forward on the same weights; afterwards every step is held to a loose band and the whole
        # trajectory to a mean |log ratio|: while the loss falls ~30 % per step a one-step-equivalent lag of 10 % of a step
        # already shows as several % (three bf16 runs of this test gave 1.9e-2, 3.4e-2 and 8.0e-2 at step 6)
        assert abs(hist[0] / ref[0] - 1) <= (1e-5 if dt == torch.float32 else 5e-3), (str(dt), hist[0], ref[0])
        logr = [abs(float(np.log(hist[it] / ref[it]))) for it in range(nsteps)]
        # run-to-run spread of these two numbers, four runs each (scripts/traj_spread.py, profiles/r03_traj_spread.txt): bf16 mean
        # 0.046-0.080 / max 0.13-0.22, fp32 mean 0.028-0.053 / max 0.07-0.13 -- with and without the fused backward alike.  That
        # was measured while the BatchNorm sums of a workgroup were accumulated in fp32 in arrival order; the step is bitwise
        # reproducible now (test_gpu_step.py::test_step_is_bitwise_reproducible), but the spread is what ANY ulp-level change of
        # the arithmetic (another box's oracle threads, a different tile assignment) does, so the thresholds stay
        assert max(logr) <= 0.30, (str(dt), logr)
        assert float(np.mean(logr)) <= (0.11 if dt == torch.bfloat16 else 0.08), (str(dt), float(np.mean(logr)))
        assert hist[-1] < 0.05 * hist[0]
        del ts, bank
