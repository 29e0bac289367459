"""Full-size parity harness: ONE training step of the HIP path at a BASELINE.json configuration (C2 400x400 [2,3,3],
C3 prostate 384x384 [2]*5, C5 512x512 [2]*4), RAM on the GPU, through the default kernel dispatch, against the oracle
run live on the same seeded inputs (reference: code/train.py:246-296 fundus, :412-465 prostate; RAM
code/dataset/fundus.py:41-61).  Used by tests/test_gpu_fullsize.py and scripts/fullsize_parity.py.

TEST INFRASTRUCTURE: imports oracle/.
"""
import random
import time

import numpy as np
import torch

CONFIGS = {
    'C2': dict(dataset='fundus', bs=[2, 3, 3], S=400),
    'C3': dict(dataset='prostate', bs=[2, 2, 2, 2, 2], S=384),
    'C5': dict(dataset='fundus', bs=[2, 2, 2, 2], S=512),
    # reference-native Fundus target-0 shape (train.py:35,541)
    'F256': dict(dataset='fundus', bs=[3, 6, 7], S=256),
    'T128': dict(dataset='fundus', bs=[2, 3, 3], S=128),
}


def rel_l2(got, ref):
    return float((got.double() - ref.double()).norm() / (ref.double().norm() + 1e-30))


def smooth_images(rng, B, S, lo, hi):
    """Band-limited random images (a fundus photograph is smooth, not white noise) plus pixel noise: low-resolution
    uniform fields bilinearly upsampled, mixed with 25 % white noise, rescaled to [lo, hi]."""
    import torch.nn.functional as F
    base = torch.from_numpy(rng.uniform(0, 1, (B, 3, S // 16, S // 16)).astype(np.float32))
    img = F.interpolate(base, size=(S, S), mode='bilinear', align_corners=False).numpy()
    img = 0.75 * img + 0.25 * rng.uniform(0, 1, (B, 3, S, S)).astype(np.float32)
    img = img.transpose(0, 2, 3, 1)
    return (lo + (hi - lo) * img).astype(np.float32)


def synth(cfg, seed=1337):
    """Inputs as the reference's datasets hold them before RAM (fundus.py:209-212 / prostate.py:177-186):
    HWC float32 source + partner, lambda = randint(1,10)/10, mask."""
    rng = np.random.RandomState(seed)
    bs, S = cfg['bs'], cfg['S']
    B = sum(bs)
    if cfg['dataset'] == 'fundus':
        src = np.round(smooth_images(rng, B, S, 0, 255))
        trg = np.round(smooth_images(rng, B, S, 0, 255))
    else:
        src = smooth_images(rng, B, S, -1, 1)
        trg = smooth_images(rng, B, S, -1, 1)
    pr = random.Random(seed)
    lam = np.array([pr.randint(1, 10) / 10 for _ in range(B)], np.float32)
    yy, xx = np.mgrid[0:S, 0:S]
    disc = np.zeros((B, S, S), bool)
    cup = np.zeros((B, S, S), bool)
    for i in range(B):
        cy, cx = rng.uniform(0.35 * S, 0.65 * S, 2)
        r_disc = rng.uniform(0.15 * S, 0.3 * S)
        r_cup = r_disc * rng.uniform(0.3, 0.7)
        d2 = (yy - cy) ** 2 + (xx - cx) ** 2
        disc[i] = d2 <= r_disc ** 2
        cup[i] = d2 <= r_cup ** 2
    if cfg['dataset'] == 'fundus':
        mask = np.stack([cup, disc], 1).astype(np.float32)            # [cup, disc] multilabel (fundus.py:227-239)
    else:
        mask = disc.astype(np.int64)
    return src, trg, lam, mask


def oracle_states(nd, K=2, seeds=(1, 2, 3)):
    from oracle import unet as OU
    return (OU.encoder_state(seed=seeds[0]), OU.decoder_state(num_classes=K, seed=seeds[1]),
            OU.rec_decoder_state(num_classes=3, num_domains=nd, seed=seeds[2]))


def oracle_ram(cfg, src, trg, lam):
    from oracle import ram as OR
    f = OR.ram_fundus if cfg['dataset'] == 'fundus' else OR.ram_prostate
    pairs = [f(src[i], trg[i], float(lam[i])) for i in range(src.shape[0])]
    return (torch.from_numpy(np.stack([p[0] for p in pairs]).astype(np.float32)),
            torch.from_numpy(np.stack([p[1] for p in pairs]).astype(np.float32)))


def oracle_step(cfg, states, img, frq, mask, consistency='kd', lr=2e-3, total_iters=21200):
    """Forward + backward of the oracle (fp32 torch CPU) -> (losses[5], rec losses, logits1, logits2, rec_soft, grads)."""
    from oracle import step as OS, unet as OU
    c = OS.StepConfig(dataset=cfg['dataset'], batch_sizes=cfg['bs'], consistency=consistency, lr=lr, total_iters=total_iters)
    e2, d2, r2 = (OU.clone_state(s, requires_grad=True) for s in states)
    t0 = time.time()
    loss, comps, inter = OS.forward_losses(e2, d2, r2, img, frq, torch.from_numpy(mask), c)
    loss.backward()
    grads = {}
    for m, sd in (('enc', e2), ('dec', d2), ('rec', r2)):
        for k in OU.param_keys(sd):
            grads[(m, k)] = sd[k].grad
    out = dict(losses=[comps[k].item() for k in ('seg1', 'dice1', 'seg2', 'dice2', 'cons')], rec=comps['rec'].tolist(),
               total=loss.item(), logit1=inter['logit1'].detach(), logit2=inter['logit2'].detach(),
               rec_soft=inter['rec_soft'].detach(), grads=grads, seconds=time.time() - t0,
               feats1=[f.detach() for f in inter['feats1']])
    return out


def hip_step(cfg, states, src, trg, lam, mask, dtype, consistency='kd', lr=2e-3, total_iters=21200, nsteps=1, graph=False,
             given_images=None):
    """One step of the product path (everything through the C ABI).  Returns the TrainStep and a dict of CPU copies.
    given_images=(img, frq) NCHW fp32: bypass RAM (used to separate RAM parity from network parity)."""
    from ramdsir import step as S
    dev = 'cuda:0'
    nd = len(cfg['bs'])
    K = 2
    bank, mods = S.make_bank(dev, 3, 16, K, nd)
    for m, sd in zip(('enc', 'dec', 'rec'), states):
        S.load_state(bank, m, sd)
    ts = S.TrainStep(bank, mods, dtype, cfg['bs'], cfg['S'], cfg['S'], dataset=cfg['dataset'], consistency=consistency,
                     lr=lr, total_iters=total_iters, ram=given_images is None, num_classes=K,
                     options=dict(side_cus=0, rec_cus=0) if graph else None)
    ts.wpack.refresh()
    T = lambda a: torch.from_numpy(a).to(dev)
    if given_images is None:
        ts.load_raw(T(src), T(trg), T(lam))
    else:
        ts.load_images(given_images[0].to(dev), given_images[1].to(dev))
    ts.load_target(T(mask))
    if graph:
        ts.capture()
    hist = []
    for _ in range(nsteps):
        ts.step()
        torch.cuda.synchronize()
        hist.append([ts.losses[i].item() for i in range(5)] + [ts.loss_dict()['loss']])
    B = sum(cfg['bs'])
    lg = ts.logits.buf.float().cpu().permute(0, 3, 1, 2)
    out = dict(losses=hist[0][:5], total=hist[0][5], hist=hist, rec=ts.rec_mse.cpu().tolist(),
               x=ts.x.buf[..., :3].float().cpu().permute(0, 3, 1, 2), logit1=lg[:B], logit2=lg[B:],
               rec_logits=ts.rec_logits.buf[..., :3].float().cpu().permute(0, 3, 1, 2),
               grads={k: bank.view(bank.grads, *k).cpu().clone() for k in bank.index})
    return ts, bank, out


def bn_shadowed_bias(key):
    """Conv biases followed by a train-mode BatchNorm: analytically zero gradient (exact zero in the HIP path, fp32
    cancellation noise in the reference); every conv except the two out1 heads."""
    return key.endswith('.bias') and '.conv' in key


def grad_table(got, ref):
    """[(rel_l2, ref_rms, module, key)] over live parameter tensors."""
    rows = []
    for (m, k), r in ref.items():
        if bn_shadowed_bias(k):
            continue
        rows.append((rel_l2(got[(m, k)], r), float(r.double().pow(2).mean().sqrt()), m, k))
    return rows
