"""-m gpu: the fused training step (all HIP, through the C ABI) against (a) the committed fixtures of the
reference's own step and (b) the oracle run live on the same inputs.  fp32 tolerances: losses 1e-4 rel,
gradients 2e-3 of each tensor's RMS (fp32 reassociation through ~40 layers), bf16: loss band only."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ramdsir import step as S, engine as E, _lib as L      # noqa: E402
S_ = S
from oracle import step as OS, unet as OU                   # noqa: E402
from golden_util import (assert_sig_close, load_step, step_states, bn_shadowed_bias, sig)   # noqa: E402

T = torch.from_numpy
DEV = 'cuda:0'

# Element-wise gradient parity of a FULL step is limited by the problem, not the kernels: ReLU masks and
# 2x2-pixel BatchNorm batches make the fp32 reference itself differ from an fp64 run of the same code by
# up to 1.8e-1 max/RMS and 1.1e-2 relative L2 on these fixtures (measured with the oracle, fp32 vs fp64).
# Full-step checks therefore use relative L2 <= 4e-2; the smooth-activation test below (slope=1, no ReLU
# kinks) holds the same composition to 2e-3, and tests/test_gpu_ops.py holds every kernel to 2e-5.
REL_L2_STEP = 4e-2
# ... and the 64x64 fixture (4x4-pixel bottleneck, BatchNorm batches of >= 32 values) to 2.5e-2 (runs give 1.2e-2 .. 1.7e-2); at the benchmark sizes
# tests/test_gpu_fullsize.py holds every gradient to the reference's own noise floor (<= 2e-2 worst, 6e-3 median).
REL_L2_BY_FIXTURE = {'fundus64': 2.5e-2, 'prostate96': 8e-2}
# prostate96: DSBN groups of ONE image at the 6x6 bottleneck = 36-sample BatchNorm batches.  The reference's own one-ulp
# noise floor there (oracle vs oracle on inputs perturbed by 6e-8, scripts/noise_floor.py's method) is 3.5-4.7e-3 median /
# 1.8e-2 worst per tensor; the HIP step sits at 3.9e-3 median and, depending on the run's accumulation order, 1.2e-2 ... 4.8e-2
# on its worst tensor (one ReLU decision of a 36-sample channel flips in about a third of the runs).  Hence a loose
# per-tensor bound AND a sharp bound on the median.
MEDIAN_REL_L2_BY_FIXTURE = {'fundus64': 1e-2, 'prostate96': 1e-2}


def rel_l2(got, ref):
    return float((got.double() - ref.double()).norm() / (ref.double().norm() + 1e-30))


def _setup(golden_dir, name, dtype, options=None):
    G, meta = load_step(golden_dir, name)
    enc, dec, rec = step_states(meta)
    nd = len(meta['batch_sizes'])
    bank, mods = S.make_bank(DEV, 3, 16, meta['num_classes'], nd)
    for m, sd in (('enc', enc), ('dec', dec), ('rec', rec)):
        S.load_state(bank, m, sd)
    ts = S.TrainStep(bank, mods, dtype, meta['batch_sizes'], meta['S'], meta['S'],
                     dataset='fundus' if name.startswith('fundus') else 'prostate', consistency=meta['consistency'],
                     lambda_rec=meta['lambda_rec'], lr=meta['base_lr'], total_iters=meta['total_iters'],
                     num_classes=meta['num_classes'], options=options)
    ts.wpack.refresh()
    return G, meta, (enc, dec, rec), bank, mods, ts


def _feed(ts, G, it):
    ts.load_images(T(G['s%d.img' % it]).to(DEV), T(G['s%d.img_freq' % it]).to(DEV))
    ts.load_target(T(G['s%d.mask' % it]).to(DEV))


@pytest.mark.parametrize('name', ['fundus', 'fundus_mse', 'prostate', 'fundus64', 'prostate96'])
def test_step_fp32_matches_reference_fixture(golden_dir, name):
    G, meta, states, bank, mods, ts = _setup(golden_dir, name, torch.float32)
    REL_L2_STEP = REL_L2_BY_FIXTURE.get(name, globals()['REL_L2_STEP'])
    _feed(ts, G, 0)
    ts.step()
    torch.cuda.synchronize()
    ld = ts.loss_dict()
    ref = G['s0.losses']
    got = [ts.losses[i].item() for i in range(5)] + [ld['loss']]
    np.testing.assert_allclose(got, ref, rtol=1e-4)
    np.testing.assert_allclose(ld['rec'], G['s0.rec_losses'], rtol=1e-4)
    # gradients of every parameter vs the reference's (signatures: L2, L1, first 8 elements)
    full = []
    for m in ('enc', 'dec', 'rec'):
        for key, shape, kind, _ in dict(mods)[m]:
            if kind != 'param':
                continue
            g = bank.g(m, key).cpu()
            if bn_shadowed_bias(key):
                assert float(g.abs().max()) == 0.0          # exact zero here, fp32 noise in the reference
                continue
            ref = G['s0.g%s.sig.%s' % (m, key)]
            np.testing.assert_allclose(float(g.double().norm()), np.sqrt(ref[2]), rtol=REL_L2_STEP, err_msg=key)
            np.testing.assert_allclose(float(g.double().abs().sum()), ref[1], rtol=REL_L2_STEP, err_msg=key)
            fk = 's0.g%s.full.%s' % (m, key)
            if fk in G.files:
                full.append(rel_l2(g, T(G[fk])))
                assert full[-1] <= REL_L2_STEP, key
    if name in MEDIAN_REL_L2_BY_FIXTURE:
        assert len(full) > 100 and float(np.median(full)) <= MEDIAN_REL_L2_BY_FIXTURE[name], (len(full), float(np.median(full)))
    # post-step state.  Adam's first update is ~lr*sign(g): elements whose gradient is below fp32 noise take
    # a platform-dependent sign, so parameters are held to |diff| <= 2*lr per element and 1e-2 on the norm;
    # running statistics / num_batches_tracked are plain fp32 averages and held to 1e-3.
    lr = meta['base_lr']
    for m in ('enc', 'dec', 'rec'):
        for key, shape, kind, _ in dict(mods)[m]:
            if bn_shadowed_bias(key):
                continue
            v = (bank.p(m, key) if kind == 'param' else bank.b(m, key)).cpu().double().reshape(-1)
            ref = G['s0.post.%s.sig.%s' % (m, key)]
            n8 = int(min(8, ref[3]))
            if kind == 'param':
                assert np.abs(v[:n8].numpy() - ref[4:4 + n8]).max() <= 2.1 * lr, key
                # a 16..64-element BatchNorm vector moves by exactly lr per element: ONE element whose gradient is noise in the
                # reference and an exact zero here (a channel dead behind its ReLU) changes the norm by 1 - sqrt(1 - 1/n) =
                # 1.6 % at n = 32 (seen once in ~10 runs of the fundus_mse fixture, convu2.bn2.bias: 0.011162 vs 0.011314)
                np.testing.assert_allclose(float(v.norm()), np.sqrt(ref[2]), rtol=1e-2 if v.numel() > 128 else 3.5e-2, err_msg=key)
            else:
                np.testing.assert_allclose(v[:n8].numpy(), ref[4:4 + n8], rtol=1e-3, atol=1e-6, err_msg=key)
    assert int(ts.iter) == 1


def test_step_fp32_three_steps_track_reference(golden_dir):
    G, meta, states, bank, mods, ts = _setup(golden_dir, 'fundus', torch.float32)
    for it in range(meta['nsteps']):
        _feed(ts, G, it)
        ts.step()
        torch.cuda.synchronize()
        np.testing.assert_allclose(ts.hyper[0].item(), G['s%d.lr_used' % it][1], rtol=1e-6)
        got = [ts.losses[i].item() for i in range(5)]
        # steps >= 1: Adam's sign-like first update amplifies fp32 noise (see tests/test_oracle_step.py)
        np.testing.assert_allclose(got, G['s%d.losses' % it][:5], rtol=1e-4 if it == 0 else 5e-2)


@pytest.mark.parametrize('name', ['fundus', 'prostate'])
def test_step_fp32_full_tensors_vs_oracle(golden_dir, name):
    """Everything the step produces, element-wise, against the oracle run here on the same inputs."""
    G, meta, (enc, dec, rec), bank, mods, ts = _setup(golden_dir, name, torch.float32)
    _feed(ts, G, 0)
    ts.step()
    torch.cuda.synchronize()
    cfg = OS.StepConfig(dataset='fundus' if name.startswith('fundus') else 'prostate', batch_sizes=meta['batch_sizes'],
                        lambda_rec=meta['lambda_rec'], consistency=meta['consistency'], lr=meta['base_lr'],
                        total_iters=meta['total_iters'], num_classes=meta['num_classes'])
    e2, d2, r2 = (OU.clone_state(s, requires_grad=True) for s in (enc, dec, rec))
    loss, comps, inter = OS.forward_losses(e2, d2, r2, T(G['s0.img']), T(G['s0.img_freq']), T(G['s0.mask']), cfg)
    loss.backward()
    B = sum(meta['batch_sizes'])
    lg = ts.logits.buf.float().cpu().permute(0, 3, 1, 2)
    for got, ref in ((lg[:B], inter['logit1']), (lg[B:], inter['logit2'])):
        rms = ref.detach().pow(2).mean().sqrt()
        assert float((got - ref.detach()).abs().max()) < 1e-4 * rms * 10
    rl = ts.rec_logits.buf.float().cpu().permute(0, 3, 1, 2)
    ref = torch.atanh(inter['rec_soft'].detach().clamp(-0.999999, 0.999999))
    assert float((torch.tanh(rl) - inter["rec_soft"].detach()).abs().max()) < 2e-3   # tiny DSBN slices (2-3 images at 2x2) amplify fp32 noise
    for m, sd in (('enc', e2), ('dec', d2), ('rec', r2)):
        for k in OU.param_keys(sd):
            g = bank.g(m, k).cpu()
            r = sd[k].grad
            if bn_shadowed_bias(k):
                continue
            assert rel_l2(g, r) <= REL_L2_STEP, (m, k, rel_l2(g, r))


def test_step_fp32_smooth_activation_tight(golden_dir):
    """Same composition with activation slope 1 (LeakyReLU(1.0) == identity): no ReLU kinks, so the whole
    fused step -- every loader mode, epilogue, BN backward, up/pool paths, losses -- must agree with the
    oracle to fp32 accuracy."""
    G, meta = load_step(golden_dir, 'fundus')
    enc, dec, rec = step_states(meta)
    bank, mods = S.make_bank(DEV, 3, 16, 2, 3)
    for m, sd in (('enc', enc), ('dec', dec), ('rec', rec)):
        S.load_state(bank, m, sd)
    ts = S.TrainStep(bank, mods, torch.float32, meta['batch_sizes'], 32, 32, dataset='fundus', consistency='kd',
                     lr=meta['base_lr'], total_iters=meta['total_iters'], slope=1.0)
    ts.wpack.refresh()
    _feed(ts, G, 0)
    ts.step()
    torch.cuda.synchronize()
    cfg = OS.StepConfig(dataset='fundus', batch_sizes=meta['batch_sizes'], consistency='kd', lr=meta['base_lr'],
                        total_iters=meta['total_iters'], slope=1.0)
    e2, d2, r2 = (OU.clone_state(s, requires_grad=True) for s in (enc, dec, rec))
    # fp32 oracle (BCE's log clamp / sigmoid saturation are defined in fp32; linear activations give large logits)
    loss, comps, inter = OS.forward_losses(e2, d2, r2, T(G['s0.img']), T(G['s0.img_freq']), T(G['s0.mask']), cfg)
    loss.backward()
    got = [ts.losses[i].item() for i in range(5)]
    np.testing.assert_allclose(got, [comps[k].item() for k in ('seg1', 'dice1', 'seg2', 'dice2', 'cons')], rtol=1e-4)
    rows = []
    for m, sd in (('enc', e2), ('dec', d2), ('rec', r2)):
        for k in OU.param_keys(sd):
            if bn_shadowed_bias(k):
                continue
            r = sd[k].grad
            rows.append((rel_l2(bank.g(m, k).cpu(), r), float(r.double().pow(2).mean().sqrt()), m, k))
    med = float(np.median([r[1] for r in rows]))
    # with linear activations a BN bias in front of a 1x1 conv + BN has an analytically zero gradient: such
    # noise-level tensors (RMS < 1e-3 of the median) are excluded from the relative check
    live = [r for r in rows if r[1] > 1e-3 * med]
    assert len(live) > 0.8 * len(rows)
    bad = [r for r in live if r[0] > 2e-3]
    assert not bad, sorted(bad, reverse=True)[:10]


def test_step_bf16_loss_band_and_graph_replay(golden_dir):
    """bf16 storage: element-wise parity with the fp32 oracle is not defined beyond single ops (SURVEY.md
    section 7); the step is held to a loss band and to bit-identical behaviour eager vs hipGraph replay
    (modulo atomics order)."""
    G, meta, states, bank, mods, ts = _setup(golden_dir, 'fundus', torch.bfloat16)
    _feed(ts, G, 0)
    ts.step()
    torch.cuda.synchronize()
    got = np.array([ts.losses[i].item() for i in range(5)])
    np.testing.assert_allclose(got, G['s0.losses'][:5], rtol=5e-2)
    # graph path
    with pytest.raises(ValueError, match='side_cus'):
        ts.capture()                                    # lane budgets are baked in: one chain would run on half of the GPU
    G2, meta2, _, bank2, mods2, ts2 = _setup(golden_dir, 'fundus', torch.bfloat16, options=dict(side_cus=0, rec_cus=0))
    _feed(ts2, G2, 0)
    ts2.capture()
    ts2.step()
    torch.cuda.synchronize()
    got2 = np.array([ts2.losses[i].item() for i in range(5)])
    np.testing.assert_allclose(got2, got, rtol=1e-2)
    assert int(ts2.iter) == 1
    rel = float((bank2.params - bank.params).abs().max())
    assert rel < 5e-3            # Adam moves every weight by ~lr; eager (lane budgets) and replay (none) group the per-workgroup sums differently


@pytest.mark.parametrize('dtype,dataset,bs,S', [(torch.float32, 'fundus', [2, 3, 3], 128), (torch.bfloat16, 'fundus', [2, 3, 3], 400),
                                                (torch.bfloat16, 'prostate', [2, 2, 2, 2, 2], 384)])
def test_step_is_bitwise_reproducible(dtype, dataset, bs, S):
    """The same parameters, optimizer state and inputs give the same BITS, run after run, on three streams: every reduction whose
    order depends on timing (waves arriving at a workgroup's BatchNorm sums, workgroups arriving at the statistic slots) accumulates
    fp32 terms in fp64, where the sum is exact and therefore order-independent WHILE the terms span less than 2^29 (csrc/conv_device.h
    flush_bstats); the weight-gradient splits are summed in a fixed order.  Round 5 (profiles/r05_determinism.txt): with other processes on the
    same GPU the step did NOT repeat -- one instruction (v_permlane32_swap in conv_small_fwd_kernel) left lane groups unswapped; fixed, and
    scripts/load_train_ab.sh is the test for that case.  Reference: train.py:608-614 (--deterministic gives cuDNN-deterministic runs there).
    Before this held (round 2) two identical fp32 runs differed by up to 8e-3 in a gradient tensor, bf16 runs by 13 %."""
    torch.manual_seed(0)
    B = sum(bs)
    bank, mods = S_.make_bank(DEV, 3, 16, 2, len(bs))
    g = torch.Generator().manual_seed(1)
    for (m, k), (off, shape) in bank.index.items():
        v = bank.p(m, k)
        if len(shape) == 4:
            v.copy_((torch.randn(shape, generator=g) * (2.0 / (shape[0] * shape[2] * shape[3])) ** 0.5).to(DEV))
        elif '.bn' in k and k.endswith('weight'):
            v.fill_(1.0)
    ts = S_.TrainStep(bank, mods, dtype, bs, S, S, dataset=dataset, consistency='kd', lr=1e-3, total_iters=100, ram=True)
    ts.wpack.refresh()
    gen = torch.Generator(device=DEV).manual_seed(5)
    if dataset == 'fundus':
        src = torch.rand(B, S, S, 3, device=DEV, generator=gen) * 255
        trg = torch.rand(B, S, S, 3, device=DEV, generator=gen) * 255
        tgt = (torch.rand(B, 2, S, S, device=DEV, generator=gen) > 0.5).float()
    else:
        src = torch.rand(B, S, S, 3, device=DEV, generator=gen) * 2 - 1
        trg = torch.rand(B, S, S, 3, device=DEV, generator=gen) * 2 - 1
        tgt = (torch.rand(B, S, S, device=DEV, generator=gen) > 0.7).long()
    lam = torch.tensor([0.1 * (1 + i % 9) for i in range(B)], device=DEV)
    ts.load_raw(src, trg, lam)
    ts.load_target(tgt)
    ts.step()                                                        # away from the initial state (Adam moments, running statistics)
    torch.cuda.synchronize()
    saved = ts._snapshot()
    runs = []
    for _ in range(3):
        ts._restore(saved)
        ts.load_raw(src, trg, lam)
        ts.load_target(tgt)
        ts.step()
        ts.step()                                                    # two steps: the second one starts from the first one's bits
        torch.cuda.synchronize()
        runs.append((bank.grads.clone(), bank.params.clone(), ts.losses.clone(), ts.rec_mse.clone(),
                     {k: v.clone() for k, v in bank.buffers.items()}))
    for r in runs[1:]:
        for a, b in zip(r[:4], runs[0][:4]):
            assert torch.equal(a, b)
        for k in r[4]:
            assert torch.equal(r[4][k], runs[0][4][k]), k
    assert float(runs[0][0].abs().max()) > 0 and torch.isfinite(runs[0][0]).all()


@pytest.mark.parametrize('fork', [1, 0])
def test_native_launch_list_equals_the_python_launch_loop(fork):
    """TrainStep.run_eager goes through rd_run_list (one C++ walk of the launch list per step: include/ramdsir.h); the instrumented
    path (Plan.run_lanes, one ctypes call per entry) must enqueue the SAME work in the same stream order: identical bits after two
    steps, on three streams and on one."""
    torch.manual_seed(0)
    bs, S = [2, 3, 3], 128
    B = sum(bs)
    bank, mods = S_.make_bank(DEV, 3, 16, 2, len(bs))
    g = torch.Generator().manual_seed(1)
    for (m, k), (off, shape) in bank.index.items():
        v = bank.p(m, k)
        if len(shape) == 4:
            v.copy_((torch.randn(shape, generator=g) * (2.0 / (shape[0] * shape[2] * shape[3])) ** 0.5).to(DEV))
        elif '.bn' in k and k.endswith('weight'):
            v.fill_(1.0)
    ts = S_.TrainStep(bank, mods, torch.bfloat16, bs, S, S, dataset='fundus', consistency='kd', lr=1e-3, total_iters=100, ram='u8',
                      options=dict(fork=fork))
    ts.wpack.refresh()
    gen = torch.Generator(device=DEV).manual_seed(5)
    src = (torch.rand(B, S, S, 3, device=DEV, generator=gen) * 255).to(torch.uint8)
    trg = (torch.rand(B, S, S, 3, device=DEV, generator=gen) * 255).to(torch.uint8)
    tgt = (torch.rand(B, 2, S, S, device=DEV, generator=gen) > 0.5).float()
    lam = torch.tensor([0.1 * (1 + i % 9) for i in range(B)], device=DEV)
    ts.load_raw(src, trg, lam)
    ts.load_target(tgt)
    torch.cuda.synchronize()
    saved = ts._snapshot()

    def python_step():
        ts.zero()
        ts.run_segment(ts.seg_a + ts.seg_b)
        ts.run_segment(ts.seg_c)

    def threaded_step():                                          # lane worker threads (rd_run_list_threads): same graph, same bits
        ts.launch_threads = True
        ts.run_eager()
        ts.launch_threads = False

    outs = []
    for stepper in (ts.run_eager, python_step, threaded_step, ts.run_eager):
        ts._restore(saved)
        stepper()
        stepper()
        torch.cuda.synchronize()
        outs.append((bank.grads.clone(), bank.params.clone(), ts.losses.clone(), ts.rec_mse.clone()))
    assert int(ts.iter) == 2
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert torch.equal(a, b)
    assert float(outs[0][0].abs().max()) > 0 and torch.isfinite(outs[0][0]).all()
    assert len(ts._native) == 1                                   # compiled once, reused


def test_stored_weight_gradient_operands_give_identical_bits():
    """TrainStep(options=dict(store_wgrad_operands=3)): the 64-wide forward / gradient launches also write act(bn(z)) / dz as they stage
    them (rd_src_t.out) and the weight gradients of those layers read the stored tensors.  The stored values are the ones the
    weight-gradient loader would have formed itself (same expression, same rounding), so gradients, parameters and losses after two
    steps are bit-identical to the default (opt-in: measured a loss in step time, ramdsir/tuning.py)."""
    torch.manual_seed(0)
    bs, S = [2, 3, 3], 256
    B = sum(bs)
    outs = []
    for store in (0, 3):
        bank, mods = S_.make_bank(DEV, 3, 16, 2, len(bs))
        g = torch.Generator().manual_seed(1)
        for (m, k), (off, shape) in bank.index.items():
            v = bank.p(m, k)
            if len(shape) == 4:
                v.copy_((torch.randn(shape, generator=g) * (2.0 / (shape[0] * shape[2] * shape[3])) ** 0.5).to(DEV))
            elif '.bn' in k and k.endswith('weight'):
                v.fill_(1.0)
        ts = S_.TrainStep(bank, mods, torch.bfloat16, bs, S, S, dataset='fundus', consistency='kd', lr=1e-3, total_iters=100, ram='u8',
                          options=dict(store_wgrad_operands=store, store_wgrad_min_c=64))
        stored = [n for n in ts.seg.nodes + ts.rec.nodes if any(t is not None for t in getattr(n, 'a_store', []))]
        assert bool(stored) == bool(store)                         # the option reaches launches of this geometry
        ts.wpack.refresh()
        gen = torch.Generator(device=DEV).manual_seed(5)
        src = (torch.rand(B, S, S, 3, device=DEV, generator=gen) * 255).to(torch.uint8)
        trg = (torch.rand(B, S, S, 3, device=DEV, generator=gen) * 255).to(torch.uint8)
        tgt = (torch.rand(B, 2, S, S, device=DEV, generator=gen) > 0.5).float()
        lam = torch.tensor([0.1 * (1 + i % 9) for i in range(B)], device=DEV)
        ts.load_raw(src, trg, lam)
        ts.load_target(tgt)
        for _ in range(2):
            ts.step()
        torch.cuda.synchronize()
        outs.append((bank.grads.clone(), bank.params.clone(), ts.losses.clone(), ts.rec_mse.clone()))
        del ts
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert float(outs[0][0].abs().max()) > 0


@pytest.mark.parametrize('dtype,dataset,bs,S', [(torch.float32, 'fundus', [2, 3, 3], 64), (torch.bfloat16, 'fundus', [2, 3, 3], 256),
                                                (torch.bfloat16, 'prostate', [2, 2, 2, 2, 2], 192)])
def test_folded_batchnorm_finalize_gives_the_bits_of_the_explicit_launches(dtype, dataset, bs, S):
    """TrainStep(options=dict(fold_finalize=...)): the default step has NO rd_bn_finalize_* launch -- the first launch that reads a
    BatchNorm's coefficients derives them from the statistic slots in its prologue (rd_src_t.fin, csrc/bn_fin.h), and the sums are
    spread over 8 slot copies instead of 64.  Same formulas, exact fp64 slot sums: gradients, parameters, losses AND every BatchNorm
    buffer (running_mean / running_var in group order for the two passes of the shared BatchNorms, num_batches_tracked; DSBN per
    domain) after three steps are bit-identical to the step with the explicit launches (nn.BatchNorm2d semantics: unet.py:17-28,
    dsbn.py:10-11)."""
    torch.manual_seed(0)
    B = sum(bs)
    outs = []
    for fold in (0, 1, 6):          # 0: explicit launches, 1: every finalize folded, 6: the default (tuning.py fold_finalize: the forward ones, and
        #                             the backward ones of the layers whose gradient and weight gradient are ONE launch)
        bank, mods = S_.make_bank(DEV, 3, 16, 2, len(bs))
        g = torch.Generator().manual_seed(1)
        for (m, k), (off, shape) in bank.index.items():
            v = bank.p(m, k)
            if len(shape) == 4:
                v.copy_((torch.randn(shape, generator=g) * (2.0 / (shape[0] * shape[2] * shape[3])) ** 0.5).to(DEV))
            elif '.bn' in k and k.endswith('weight'):
                v.copy_((1.0 + 0.1 * torch.randn(shape, generator=g)).to(DEV))
            elif '.bn' in k and k.endswith('bias'):
                v.copy_((0.1 * torch.randn(shape, generator=g)).to(DEV))
        ts = S_.TrainStep(bank, mods, dtype, bs, S, S, dataset=dataset, consistency='kd', lr=1e-3, total_iters=100, ram=True,
                          options=dict(fold_finalize=fold))
        names = [op[0].__name__ for op in ts._ops if op[0] is not None]
        n_fin = sum(1 for nm in names if nm.startswith('rd_bn_finalize'))
        assert {0: n_fin > 70, 1: n_fin == 0, 6: 0 < n_fin <= 38}[fold], n_fin         # 38 + 38 sites (fp32 has no one-launch backward: 6 keeps all 38 backward launches)
        ts.wpack.refresh()
        gen = torch.Generator(device=DEV).manual_seed(5)
        if dataset == 'fundus':
            src = torch.rand(B, S, S, 3, device=DEV, generator=gen) * 255
            trg = torch.rand(B, S, S, 3, device=DEV, generator=gen) * 255
            tgt = (torch.rand(B, 2, S, S, device=DEV, generator=gen) > 0.5).float()
        else:
            src = torch.rand(B, S, S, 3, device=DEV, generator=gen) * 2 - 1
            trg = torch.rand(B, S, S, 3, device=DEV, generator=gen) * 2 - 1
            tgt = (torch.rand(B, S, S, device=DEV, generator=gen) > 0.7).long()
        lam = torch.tensor([0.1 * (1 + i % 9) for i in range(B)], device=DEV)
        ts.load_raw(src, trg, lam)
        ts.load_target(tgt)
        for _ in range(3):
            ts.step()
        torch.cuda.synchronize()
        outs.append((bank.grads.clone(), bank.params.clone(), ts.losses.clone(), ts.rec_mse.clone(),
                     {k: v.clone() for k, v in bank.buffers.items()}))
        del ts
    for other in outs[1:]:
        for a, b in zip(outs[0][:4], other[:4]):
            assert torch.equal(a, b)
        for k, v in outs[0][4].items():
            assert torch.equal(v, other[4][k]), k
    nbt = [v for (m, k), v in outs[1][4].items() if k.endswith('num_batches_tracked')]
    assert {int(v) for (m, k), v in outs[1][4].items() if k.endswith('num_batches_tracked') and m != 'rec'} == {6}    # two passes x three steps
    assert {int(v) for (m, k), v in outs[1][4].items() if k.endswith('num_batches_tracked') and m == 'rec'} == {3}
    assert float(outs[0][0].abs().max()) > 0 and torch.isfinite(outs[0][0]).all()


@pytest.mark.parametrize('dataset', ['fundus', 'prostate'])
def test_pipelined_ram_steps_equal_classical_steps(dataset):
    """TrainStep.load_raw_next(): the NEXT batch is uploaded into the other input slot and mixed (RAM) into that slot's copy of the
    network input on the restoration lane while the current step's encoder backward runs, instead of at the head of its own step (the reference mixes per sample in DataLoader workers,
    code/dataset/fundus.py:197-225, so its step never waits for RAM either).  Three different batches, classical
    (load_raw / load_target / step) against pipelined (load_raw_next before the step, load_target after it): identical bits in
    losses, gradients, parameters and the mixed network input."""
    torch.manual_seed(0)
    bs, S = [2, 3, 3], 64
    B = sum(bs)
    bank, mods = S_.make_bank(DEV, 3, 16, 2, len(bs))
    g = torch.Generator().manual_seed(1)
    for (m, k), (off, shape) in bank.index.items():
        v = bank.p(m, k)
        if len(shape) == 4:
            v.copy_((torch.randn(shape, generator=g) * (2.0 / (shape[0] * shape[2] * shape[3])) ** 0.5).to(DEV))
        elif '.bn' in k and k.endswith('weight'):
            v.fill_(1.0)
    ts = S_.TrainStep(bank, mods, torch.float32, bs, S, S, dataset=dataset, consistency='kd', lr=1e-3, total_iters=100,
                      ram='u8' if dataset == 'fundus' else True)
    ts.wpack.refresh()
    gen = torch.Generator(device=DEV).manual_seed(5)
    batches = []
    for i in range(3):
        if dataset == 'fundus':
            src = (torch.rand(B, S, S, 3, device=DEV, generator=gen) * 255).to(torch.uint8)
            trg = (torch.rand(B, S, S, 3, device=DEV, generator=gen) * 255).to(torch.uint8)
            tgt = (torch.rand(B, 2, S, S, device=DEV, generator=gen) > 0.5).float()
        else:
            src = torch.rand(B, S, S, 3, device=DEV, generator=gen) * 2 - 1
            trg = torch.rand(B, S, S, 3, device=DEV, generator=gen) * 2 - 1
            tgt = (torch.rand(B, S, S, device=DEV, generator=gen) > 0.7).long()
        lam = torch.tensor([0.1 * (1 + (i + j) % 9) for j in range(B)], device=DEV)
        batches.append((src, trg, lam, tgt))
    torch.cuda.synchronize()
    saved = ts._snapshot()

    def collect():
        torch.cuda.synchronize()
        return (ts.losses.clone(), ts.rec_mse.clone(), bank.grads.clone(), bank.params.clone(), ts.x_current().clone())

    # classical
    ref = []
    for src, trg, lam, tgt in batches:
        ts.load_raw(src, trg, lam)
        ts.load_target(tgt)
        ts.step()
        ref.append(collect())
    # pipelined: after step i the current slot's x already holds batch i+1 (mixed during step i), so it is compared one step later
    ts._restore(saved)
    ts.load_raw(*batches[0][:3])
    ts.load_target(batches[0][3])
    got = []
    for i in range(3):
        if i + 1 < 3:
            ts.load_raw_next(*batches[i + 1][:3])
        ts.step()
        if i + 1 < 3:
            ts.load_target(batches[i + 1][3])
        got.append(collect())
    assert int(ts.iter) == 3
    for i in range(3):
        for a, b in zip(got[i][:4], ref[i][:4]):
            assert torch.equal(a, b), i
    assert torch.equal(got[0][4], ref[1][4]) and torch.equal(got[1][4], ref[2][4]) and torch.equal(got[2][4], ref[2][4])
    assert not torch.equal(ref[0][4], ref[1][4])
    # a classical step after pipelined ones (load_raw resets the slot state) is again the classical result
    ts._restore(saved)
    ts.load_raw(*batches[0][:3])
    ts.load_target(batches[0][3])
    ts.step()
    again = collect()
    for a, b in zip(again, ref[0]):
        assert torch.equal(a, b)


def test_pipelined_data_parallel_step_equals_pipelined_plain_step():
    """The data-parallel step (ramdsir/ddp.py: segments A / B1 / B2 / C through the native launch list, three bucket exchanges) with
    the next batch's RAM forked beside segment B1, on a one-rank RCCL group: identical bits to the single-process pipelined step."""
    import torch.distributed as dist
    from ramdsir import ddp as D
    created = not dist.is_initialized()
    if created:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29547')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        bs, S = [2, 3, 3], 64
        B = sum(bs)
        gen = torch.Generator(device=DEV).manual_seed(9)
        batches = []
        for i in range(3):
            src = (torch.rand(B, S, S, 3, device=DEV, generator=gen) * 255).to(torch.uint8)
            trg = (torch.rand(B, S, S, 3, device=DEV, generator=gen) * 255).to(torch.uint8)
            tgt = (torch.rand(B, 2, S, S, device=DEV, generator=gen) > 0.5).float()
            lam = torch.tensor([0.1 * (1 + (i + j) % 9) for j in range(B)], device=DEV)
            batches.append((src, trg, lam, tgt))
        outs = []
        for use_ddp in (False, True):
            torch.manual_seed(0)
            bank, mods = S_.make_bank(DEV, 3, 16, 2, len(bs))
            g = torch.Generator().manual_seed(1)
            for (m, k), (off, shape) in bank.index.items():
                v = bank.p(m, k)
                if len(shape) == 4:
                    v.copy_((torch.randn(shape, generator=g) * (2.0 / (shape[0] * shape[2] * shape[3])) ** 0.5).to(DEV))
                elif '.bn' in k and k.endswith('weight'):
                    v.fill_(1.0)
            ts = S_.TrainStep(bank, mods, torch.bfloat16, bs, S, S, dataset='fundus', consistency='kd', lr=1e-3, total_iters=100, ram='u8')
            ts.wpack.refresh()
            stepper = D.DataParallelStep(ts).step if use_ddp else ts.step
            ts.load_raw(*batches[0][:3])
            ts.load_target(batches[0][3])
            for i in range(3):
                if i + 1 < 3:
                    ts.load_raw_next(*batches[i + 1][:3])
                stepper()
                if i + 1 < 3:
                    ts.load_target(batches[i + 1][3])
            torch.cuda.synchronize()
            assert int(ts.iter) == 3
            outs.append((bank.params.clone(), bank.grads.clone(), ts.losses.clone(), ts.rec_mse.clone()))
        for a, b in zip(outs[0], outs[1]):
            assert torch.equal(a, b)
        assert torch.isfinite(outs[0][0]).all() and float(outs[0][1].abs().max()) > 0
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize('dataset,bs,S', [('fundus', [2, 3, 3], 400), ('prostate', [2, 2, 2, 2, 2], 384), ('fundus', [2, 2, 2, 2], 512)])
def test_step_at_baseline_config_shapes(dataset, bs, S):
    """BASELINE.json configs 1/3/5 at full size (bf16, hipGraph, RAM on the GPU): size-independent properties --
    finite losses, RAM output in range and identity for lambda=1, loss decreasing on a fixed batch, running
    statistics tracked twice per step for the shared BNs and once per step for each domain's DSBN."""
    torch.manual_seed(0)
    B = sum(bs)
    K = 2
    bank, mods = S_.make_bank(DEV, 3, 16, K, len(bs))
    g = torch.Generator().manual_seed(1)
    for (m, k), (off, shape) in bank.index.items():
        v = bank.p(m, k)
        if len(shape) == 4:
            v.copy_((torch.randn(shape, generator=g) * (2.0 / (shape[0] * shape[2] * shape[3])) ** 0.5).to(DEV))
        elif '.bn' in k and k.endswith('weight'):
            v.fill_(1.0)
    ts = S_.TrainStep(bank, mods, torch.bfloat16, bs, S, S, dataset=dataset, consistency='kd', lr=1e-3, total_iters=100, ram=True,
                      options=dict(side_cus=0, rec_cus=0))           # captured below: one chain, no lane budgets
    ts.wpack.refresh()
    if dataset == 'fundus':
        src = torch.rand(B, S, S, 3, device=DEV) * 255
        trg = torch.rand(B, S, S, 3, device=DEV) * 255
        tgt = (torch.rand(B, 2, S, S, device=DEV) > 0.5).float()
    else:
        src = torch.rand(B, S, S, 3, device=DEV) * 2 - 1
        trg = torch.rand(B, S, S, 3, device=DEV) * 2 - 1
        tgt = (torch.rand(B, S, S, device=DEV) > 0.7).long()
    lam = torch.tensor([1.0] + [0.1 * (1 + i % 9) for i in range(B - 1)], device=DEV)
    ts.load_raw(src, trg, lam)
    ts.load_target(tgt)
    ts.capture()
    hist = []
    for _ in range(6):
        ts.step()
        hist.append(ts.loss_dict()['loss'])
    assert all(np.isfinite(hist)), hist
    assert hist[-1] < hist[0], hist
    x = ts.x.buf[..., :ts.c].float()
    assert float(x.min()) >= -1.0 and float(x.max()) <= 1.0
    assert torch.equal(x[0], x[B])                                  # lambda = 1: img_freq == img
    assert int(bank.b('enc', 'convd1.bn1.num_batches_tracked')) == 12
    assert int(bank.b('rec', 'convu1.bn3.bns.%d.num_batches_tracked' % (len(bs) - 1))) == 6
    assert int(ts.iter) == 6


def test_data_parallel_step_matches_plain_step_on_one_rank(golden_dir):
    """ramdsir/ddp.py on a world of ONE rank (RCCL all-reduce = identity): the segmented launch (A / B1 / B2 / C with the
    three bucket exchanges on the communication stream, lanes left open across segment boundaries) must produce the same
    gradients and losses as the plain step, eagerly and through the per-segment hipGraphs; the buckets must tile the
    arena.  Gradients are compared after the FIRST step at the tolerance of the other full-step tests (4e-2 relative L2):
    on this 32x32 fixture two PLAIN runs already differ by 6e-3 (1.2e-2 in the restoration decoder) through the atomics'
    summation order amplified by 2x2-pixel BatchNorm batches and ReLU kinks (measured; DESIGN.md 'Numerics'), and by 8e-3
    in the parameters after two Adam steps, so the second step is only held to the loss band."""
    import torch.distributed as dist
    from ramdsir import ddp as D
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29577')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        G, meta, states, bank, mods, ts = _setup(golden_dir, 'fundus', torch.float32)
        _feed(ts, G, 0)
        ts.step()
        torch.cuda.synchronize()
        g_ref, l_ref = bank.grads.cpu().clone(), [ts.losses[i].item() for i in range(5)]
        _feed(ts, G, 1)
        ts.step()
        torch.cuda.synchronize()
        l_ref2 = [ts.losses[i].item() for i in range(5)]
        for use_graph, own_comm in ((False, False), (True, False), (False, True)):
            G2, _, _, bank2, _, ts2 = _setup(golden_dir, 'fundus', torch.float32, options=dict(ddp_own_comm_stream=own_comm))
            runner = D.DataParallelStep(ts2)
            # RCCL: the asynchronous AVG branch of GradBuckets.reduce (gloo, tests/test_gpu_ddp.py, takes the SUM + divide one);
            # the collectives are launched from the weight-gradient lane unless a stream of their own is asked for
            assert runner.buckets._avg == dist.ReduceOp.AVG and (runner.comm is ts2.side[0]) == (not own_comm)
            assert runner.buckets.reduce(1, async_op=True) is not None
            b = runner.buckets.bounds
            assert b[0] == 0 and b[-1] == bank2.n and b == sorted(b) and len(b) == 4
            assert b[2] == bank2.module_range['enc'][1] and 0 < b[1] < b[2]            # encoder levels 1-2 | 3-5 | decoders
            if use_graph:
                runner.capture()
            _feed(ts2, G2, 0)
            runner.step()
            torch.cuda.synchronize()
            np.testing.assert_allclose([ts2.losses[i].item() for i in range(5)], l_ref, rtol=1e-5)
            assert rel_l2(bank2.grads.cpu(), g_ref) < 4e-2, (use_graph, own_comm, rel_l2(bank2.grads.cpu(), g_ref))
            assert float(bank2.grads.abs().max()) > 0
            _feed(ts2, G2, 1)
            runner.step()
            torch.cuda.synchronize()
            assert int(ts2.iter) == 2
            np.testing.assert_allclose([ts2.losses[i].item() for i in range(5)], l_ref2, rtol=1e-2)
    finally:
        if created:
            dist.destroy_process_group()


def test_lane_forks_take_their_event_from_the_producing_launch():
    """rd_run_list_bind_fork_events (include/ramdsir.h): a lane fork that directly follows a launch of the main stream waits for an
    event bound to that launch's own dispatch packet (hipExtLaunchKernelGGL stop event) instead of a hipEventRecord behind it -- the
    same dependency without a packet of its own on the main stream (scripts/probe/ext_event.hip: 3 us per fork; the 128 x 128 step
    1.69 -> 1.63 ms).  The step's forks are served that way, and the step gives the bits of the recorded-event walk."""
    import ctypes
    from ramdsir import _lib as L_
    lib = L_.lib()

    def counts():
        b, r = ctypes.c_longlong(0), ctypes.c_longlong(0)
        lib.rd_run_list_fork_counts(ctypes.byref(b), ctypes.byref(r))
        return b.value, r.value

    out = {}
    try:
        for mode in (0, 1):
            lib.rd_run_list_bind_fork_events(mode)
            torch.manual_seed(0)
            bank, mods = S_.make_bank(DEV, 3, 16, 2, 3)
            g = torch.Generator().manual_seed(1)
            for (m, k), (off, shape) in bank.index.items():
                v = bank.p(m, k)
                if len(shape) == 4:
                    v.copy_((torch.randn(shape, generator=g) * (2.0 / (shape[0] * shape[2] * shape[3])) ** 0.5).to(DEV))
                elif '.bn' in k and k.endswith('weight'):
                    v.fill_(1.0)
            ts = S_.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], 64, 64, dataset='fundus', consistency='kd', lr=2e-3, total_iters=100, ram='u8')
            ts.wpack.refresh()
            gen = torch.Generator(device=DEV).manual_seed(5)
            src = (torch.rand(8, 64, 64, 3, device=DEV, generator=gen) * 255).to(torch.uint8)
            trg = (torch.rand(8, 64, 64, 3, device=DEV, generator=gen) * 255).to(torch.uint8)
            ts.load_raw(src, trg, torch.tensor([0.1 * (1 + j) for j in range(8)], device=DEV))
            ts.load_target((torch.rand(8, 2, 64, 64, device=DEV, generator=gen) > 0.5).float())
            c0 = counts()
            for _ in range(4):
                ts.step()
            torch.cuda.synchronize()
            c1 = counts()
            out[mode] = (bank.params.clone(), bank.grads.clone(), ts.losses.clone(), c1[0] - c0[0], c1[1] - c0[1])
    finally:
        lib.rd_run_list_bind_fork_events(1)
    assert out[0][3] == 0 and out[0][4] > 0, out[0][3:]                  # recorded events only
    assert out[1][3] > 0 and out[1][3] >= 0.9 * out[0][4], (out[1][3:], out[0][3:])   # (nearly) every fork follows a launch directly
    for a, b in zip(out[0][:3], out[1][:3]):
        assert torch.equal(a.reshape(-1).view(torch.uint8), b.reshape(-1).view(torch.uint8))


@pytest.mark.parametrize('reps,side', [(300, 64), (100, 400)])
def test_step_repeats_bit_for_bit_while_other_processes_share_the_gpu(reps, side):
    """Three processes at once, each repeating ONE step `reps` times from the same saved state (scripts/step_repeat_stress.py) and
    comparing parameters, gradients, losses and statistics with its first repetition: 0 differences.  Until round 5 this failed in 1-2 % of
    the repetitions -- conv_small_fwd_kernel's v_permlane32_swap left 16-lane groups unswapped when the GPU was time-sliced between
    processes: 16 pixels holding the bias (profiles/r05_determinism.txt) -- while every single-process test passed.  Round 6 found the
    real cause (a packed fp32 add with op_sel behind the regroup reads 0 beside another kernel's MFMAs: profiles/r06_pk_opsel_erratum.txt; the
    library is built without the vectorizers that form such instructions), and the case at the BENCH shape (C2: 400 x 400, bf16, [2, 3, 3]) puts every
    kernel family of the step under the same load: conv_ws_kernel's multi-tile ranges, conv_pf_kernel at two workgroups per CU, the
    fused backward, the small-channel gradient launches, the 128-wide weight gradients."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'scripts', 'step_repeat_stress.py'), str(reps), str(side)]
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for _ in range(3)]
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-1500:]
        assert '%d repetitions of one %dx%d step: 0 differ from the first' % (reps, side, side) in o, o[-1500:]


def test_pipelined_ram_output_equals_ram_alone_bit_for_bit():
    """The pipelined step mixes the next batch (RAM) on the restoration lane BESIDE the encoder backward.  Until round 6 the mixed input then
    differed from RAM run alone in 44 % of the steps (single elements of a transform off by a few per cent = rows / columns of img_freq off
    by a bf16 ulp or two): the Stockham kernels produced wrong results whenever workgroups of certain other kernels shared their CU
    (profiles/r06_ram_coresidency.txt).  They claim the whole LDS of their CU now; 300 pipelined steps at the bench shape must give the
    bits of RAM alone (scripts/r6/pipelined_x_check.py; reference: source_to_target_freq, code/dataset/fundus.py:41-61)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'r6', 'pipelined_x_check.py'), '300'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-1500:]
    assert '300 pipelined steps: the mixed input differs from RAM alone in 0' in out, out[-1500:]


@pytest.mark.parametrize('family', ['rd_conv conv_kernel', 'rd_conv conv_small_kernel'])
def test_ram_beside_conv_launches_on_another_stream(family):
    """rd_ram_mix repeated on one stream while the step's conv launches of one family run on a second stream of the same process
    (scripts/r6/ram_stress.py): kept row bins, column results and both outputs bit-identical in every repetition (before the whole-CU LDS
    claim: 477 / 553 of 600)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RAM_STRESS_INPROC=family, RAM_STRESS_AGG_REPEAT='2')
    r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'r6', 'ram_stress.py'), '150', '4'], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-1500:]
    assert '150 repetitions of rd_ram_mix (burst 4): 0 differ from the first' in out, out[-1500:]
