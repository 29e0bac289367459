"""-m gpu: synthetic-domain Dice proxy of BASELINE.json's "Dice vs reference on held-out domain 0" (tests/dice_proxy.py).
300 iterations at 64x64, batch 8 = [2,3,3] over three source domains, same initial weights and batch stream for every
run; held-out domain 0 evaluated as code/train.py:91-132 does.

What was measured.  Round 3 (profiles/r03_dice_proxy.json; before the step became bitwise reproducible, so repeated HIP runs were
different draws): fp32 oracle 90.0 / 89.4 / 89.4 by host; HIP fp32 mean 90.2, sd 0.8 (9 runs); oracle under the bf16 rounding model
(oracle.unet.rounding) 91.2 / 89.9 / 89.9; HIP bf16 mean 92.6, sd 1.7 (9 runs) -- 2.5 points above the rounding model, unexplained then.
Round 5 (the step is deterministic now, so the spread is probed the way the reference's own is: initial parameters x (1 + 1e-6 N(0,1)),
the SAME perturbation seeds for every trainer; scripts/dice_proxy_oracle_spread.py on the CPU, scripts/dice_proxy_hip_spread.py on the
GPU; profiles/r05_dice_proxy_oracle_spread.json, profiles/r05_dice_proxy_hip_spread.json):
  oracle, bf16 rounding model   mean 90.7, sd 0.7 (8 runs)        HIP bf16   mean 91.5, sd 1.2 (8 runs; the unperturbed run: 92.79)
  oracle, fp32                  mean 90.1, sd 0.6 (8 runs)        HIP fp32   mean 90.3, sd 0.9 (8 runs; the unperturbed run: 90.21)
The bf16 "gap" is 0.8 +- 0.5 points between the two DISTRIBUTIONS (1.7 sigma; fp32: 0.2 +- 0.4) -- the 2.5 of round 3 compared nine noisy HIP draws, whose
unperturbed member happens to be a high one, with two or three oracle runs.  What remains is a wider spread of the bf16 HIP step (1.2
against 0.7), not a shift that the rounding points fail to explain.
The gates, two-sided.  fp32: one HIP run against the one oracle run, 3.5 points (3 sigma of the difference of two draws, sqrt(0.9^2 +
0.6^2) = 1.1, + the 0.2 between the means).  bf16: every kernel edit that moves an ulp re-draws the HIP run, and ONE draw against ONE
oracle draw at 3.5 points = (3.5 - 0.8) / 1.4 = 1.9 sigma failed one edit in forty by construction (it did: 93.45 against 89.94, 3.51).
So the bf16 gate is on the DISTRIBUTION: the mean of three HIP draws (the unperturbed start and perturbation seeds 0 and 1 of the
profile above) against the committed mean of the eight oracle draws under the rounding model (profiles/r05_dice_proxy_oracle_spread.json:
90.7), within 0.8 + 3 sqrt(1.2^2 / 3 + 0.7^2 / 8) = 3.0 points; a single draw only has to stay within 5 of the single oracle-model run
(0.8 + 3 x 1.4)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import dice_proxy as DP                                     # noqa: E402

N_ITERS, REPS = 300, 2


@pytest.fixture(scope='module')
def runs():
    from oracle import unet as OU
    train, test = DP.make_data()
    stream = DP.batch_stream(train, N_ITERS)
    out = {'test': test, 'oracle': [DP.train_oracle(stream)]}
    with OU.rounding(torch.bfloat16):
        out['oracle_bf16_model'] = [DP.train_oracle(stream)]
    out['hip_f32'] = [DP.train_hip(stream, torch.float32) for _ in range(REPS)]
    out['hip_bf16'] = [DP.train_hip(stream, torch.bfloat16) for _ in range(REPS)]
    out['hip_bf16_draws'] = [DP.train_hip(stream, torch.bfloat16, perturb_seed=sd) for sd in (0, 1)]
    return out


def _avg(states, test):
    c, d = DP.evaluate_with_oracle(states, test)
    return 50.0 * (c + d)                                    # train.py:132: (cup + disc) * 100 / 2


def test_runs_learn_the_task_and_agree_on_held_out_dice(runs):
    test = runs['test']
    avg = {k: [_avg(st, test) for st, _ in runs[k]] for k in ('oracle', 'oracle_bf16_model', 'hip_f32', 'hip_bf16')}
    print('held-out domain 0, average Dice x100:', {k: [round(v, 2) for v in vs] for k, vs in avg.items()})
    for k in avg:
        for st, hist in runs[k]:
            assert all(np.isfinite(hist)) and hist[-1] < 0.1 * hist[0], (k, hist[0], hist[-1])
    ref, ref_b = avg['oracle'][0], avg['oracle_bf16_model'][0]
    assert ref > 85.0, avg                                   # the task is learnt: the comparison is not about noise
    for k in ('hip_f32', 'hip_bf16'):                        # 300 steps from the same weights and batches: the same bits
        for sa, sb in zip(runs[k][0][0], runs[k][1][0]):     # (encoder, decoder) state dicts of run 0 / run 1
            for key in sa:
                assert torch.equal(sa[key], sb[key]), (k, key)
    assert max(abs(v - ref) for v in avg['hip_f32']) <= 3.5, avg     # fp32 kernels vs the fp32 reference arithmetic (sd of a draw: 0.9)
    # bf16 kernels vs the oracle with the same rounding points, TWO-SIDED, on the distribution (docstring): three draws against the
    # committed mean of the oracle's eight
    import json
    import os
    spread = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r05_dice_proxy_oracle_spread.json')))
    model_mean = float(spread['oracle_bf16_rounding_model']['mean'])
    draws = [avg['hip_bf16'][0]] + [_avg(st, test) for st, _ in runs['hip_bf16_draws']]
    print('HIP bf16 draws:', [round(v, 2) for v in draws], 'mean %.2f against the oracle-model distribution mean %.2f' % (np.mean(draws), model_mean))
    assert abs(np.mean(draws) - model_mean) <= 3.0, (draws, model_mean)
    assert max(abs(v - ref_b) for v in draws) <= 5.0, (draws, avg)
    assert np.mean(draws) >= ref - 3.0, (draws, avg)         # ... and the bench dtype is not worse than the reference's fp32
    assert ref_b >= ref - 2.5, avg                           # nor is the oracle with the same rounding points


def test_product_evaluation_of_the_trained_weights_matches_the_oracle_evaluation(runs):
    """The trained HIP weights evaluated by the drop-in modules (HIP eval-mode forward, what train.py::test_fundus runs) and by
    the oracle forward give the same Dice."""
    test = runs['test']
    for k in ('hip_f32', 'hip_bf16'):
        st = runs[k][0][0]
        a = DP.evaluate_with_oracle(st, test)
        b = DP.evaluate_with_product(st, test)
        assert abs(a[0] - b[0]) <= 2e-3 and abs(a[1] - b[1]) <= 2e-3, (k, a, b)
