"""Synthetic-domain Dice proxy (TEST INFRASTRUCTURE; imports oracle/).

BASELINE.json's second metric is "Dice vs reference on held-out domain 0"; the Fundus data are not redistributable
(/root/reference/README.md:21-25) and absent.  The closest thing this box allows: a four-domain synthetic Fundus-like
task -- a bright disc with a brighter cup inside on a smooth textured background, each domain with its own colour cast,
contrast and noise level -- trained for a few hundred iterations with the reference's recipe (code/train.py:225-296: RAM with
an out-of-domain partner, seg on img and img_freq, KD consistency, per-domain restoration, Adam + poly LR) from the SAME
initial weights on the SAME batch stream by (i) the oracle on the CPU in fp32, (ii) the HIP step in fp32, (iii) the HIP step
in bf16, then evaluated on the held-out domain 0 as train.py:91-132 does (BatchNorm in eval mode, threshold 0.75, largest
component + hole filling, Dice with +1 smoothing: utils/metrics.py, pinned to the reference's own output)."""
import random

import numpy as np
import torch
from PIL import Image

S = 64
BATCH = [2, 3, 3]                 # domains 1, 2, 3 (train.py:35-38 pattern: one entry per source domain)
N_TRAIN, N_TEST = 24, 48
CUP_RATIO, CUP_GAIN = (0.5, 0.7), 0.9        # cup radius / disc radius; cup contrast relative to the disc's

# per-domain appearance: (colour gain, background level, disc contrast, noise sigma)
DOMAINS = [((1.00, 0.85, 0.70), 70, 70, 10.0),     # 0: held out
           ((0.95, 0.90, 0.80), 90, 60, 6.0),
           ((0.80, 1.00, 0.85), 60, 80, 14.0),
           ((1.00, 0.75, 0.95), 80, 55, 9.0)]


def _image(rng, d):
    gain, bg, contrast, sigma = DOMAINS[d]
    low = rng.uniform(-1, 1, (4, 4)).astype(np.float32)
    tex = np.array(Image.fromarray(low).resize((S, S), Image.BILINEAR))
    yy, xx = np.mgrid[0:S, 0:S]
    cy, cx = rng.uniform(0.35 * S, 0.65 * S, 2)
    r_disc = rng.uniform(0.18, 0.30) * S
    r_cup = r_disc * rng.uniform(*CUP_RATIO)
    dist = np.sqrt((yy - cy) ** 2 + (xx - cx) ** 2)
    disc, cup = dist <= r_disc, dist <= r_cup
    soft = lambda r: 1.0 / (1.0 + np.exp((dist - r) / 1.2))             # anti-aliased edges
    lum = bg + 25.0 * tex + contrast * soft(r_disc) + CUP_GAIN * contrast * soft(r_cup)
    img = lum[..., None] * np.array(gain, np.float32)[None, None, :] + rng.normal(0, sigma, (S, S, 3))
    return np.clip(np.round(img), 0, 255).astype(np.uint8), np.stack([cup, disc], 0).astype(np.float32)


def make_data(seed=4242):
    rng = np.random.RandomState(seed)
    train = {d: [_image(rng, d) for _ in range(N_TRAIN)] for d in (1, 2, 3)}
    test = [_image(rng, 0) for _ in range(N_TEST)]
    return train, test


def batch_stream(train, n_iters, seed=99):
    """[(src uint8 [B,S,S,3], trg uint8, lam [B], mask [B,2,S,S])] -- every draw made once, shared by all three runs: images in a
    per-domain shuffled order, the RAM partner from another SOURCE domain (is_out_domain, fundus.py:201-208), lambda =
    randint(1,10)/10 (fundus.py:35)."""
    rng = np.random.RandomState(seed)
    pr = random.Random(seed)
    order = {d: [] for d in train}
    out = []
    for _ in range(n_iters):
        src, trg, lam, msk = [], [], [], []
        for d, b in zip((1, 2, 3), BATCH):
            for _ in range(b):
                if not order[d]:
                    order[d] = list(rng.permutation(len(train[d])))
                i = order[d].pop()
                other = int(rng.choice([e for e in (1, 2, 3) if e != d]))
                j = int(rng.randint(len(train[other])))
                src.append(train[d][i][0])
                msk.append(train[d][i][1])
                trg.append(train[other][j][0])
                lam.append(pr.randint(1, 10) / 10)
        out.append((np.stack(src), np.stack(trg), np.array(lam, np.float32), np.stack(msk)))
    return out


def initial_states():
    from oracle import unet as OU
    return (OU.encoder_state(seed=21), OU.decoder_state(num_classes=2, seed=22), OU.rec_decoder_state(num_classes=3, num_domains=3, seed=23))


def train_oracle(stream, lr=2e-3, perturb_seed=None):
    """perturb_seed: the initial conv weights and BatchNorm parameters multiplied by (1 + 1e-6 N(0, 1)) -- a one-ulp-sized perturbation
    of the same start: what the spread of ONE recipe over 300 Adam steps is (the ReLU decisions amplify it, DESIGN.md numerics)."""
    from oracle import ram as OR, step as OS, unet as OU
    torch.set_num_threads(min(torch.get_num_threads(), 16))      # 64x64 convs: more threads than that only add overhead
    enc, dec, rec = (OU.clone_state(s) for s in initial_states())
    if perturb_seed is not None:
        g = torch.Generator().manual_seed(int(perturb_seed))
        for sd in (enc, dec, rec):
            for k in OU.param_keys(sd):
                sd[k].mul_(1.0 + 1e-6 * torch.randn(sd[k].shape, generator=g))
    opt = {m: OS.adam_state({k: sd[k] for k in OU.param_keys(sd)}) for m, sd in (('enc', enc), ('dec', dec), ('rec', rec))}
    cfg = OS.StepConfig(dataset='fundus', batch_sizes=BATCH, consistency='kd', lr=lr, total_iters=len(stream))
    hist = []
    for it, (src, trg, lam, msk) in enumerate(stream):
        pairs = [OR.ram_fundus(src[i].astype(np.float32), trg[i].astype(np.float32), float(lam[i])) for i in range(src.shape[0])]
        img = torch.from_numpy(np.stack([p[0] for p in pairs]).astype(np.float32))
        frq = torch.from_numpy(np.stack([p[1] for p in pairs]).astype(np.float32))
        comps, _ = OS.train_step(enc, dec, rec, opt, img, frq, torch.from_numpy(msk), cfg, it)
        hist.append(comps['total'].item())
    return (enc, dec), hist


def train_hip(stream, dtype, lr=2e-3, device='cuda:0', perturb_seed=None, options=None):
    """perturb_seed: as in train_oracle (the same perturbed start for the same seed)."""
    from ramdsir import step as S_
    from ramdsir import engine as E
    from oracle import unet as OU
    bank, mods = S_.make_bank(device, 3, 16, 2, 3)
    states = [OU.clone_state(sd) for sd in initial_states()]
    if perturb_seed is not None:
        g = torch.Generator().manual_seed(int(perturb_seed))
        for sd in states:
            for k in OU.param_keys(sd):
                sd[k].mul_(1.0 + 1e-6 * torch.randn(sd[k].shape, generator=g))
    for m, sd in zip(('enc', 'dec', 'rec'), states):
        S_.load_state(bank, m, sd)
    ts = S_.TrainStep(bank, mods, dtype, BATCH, S, S, dataset='fundus', consistency='kd', lr=lr, total_iters=len(stream), ram='u8', options=options)
    ts.wpack.refresh()
    hist = []
    for src, trg, lam, msk in stream:
        ts.load_raw(torch.from_numpy(src).to(device), torch.from_numpy(trg).to(device), torch.from_numpy(lam).to(device))
        ts.load_target(torch.from_numpy(msk).to(device))
        ts.step()
        hist.append(ts.loss_dict()['loss'])
    torch.cuda.synchronize()
    enc = {k: v.cpu() for k, v in S_.state_dict_of(bank, 'enc', E.encoder_specs(3, 16)).items()}
    dec = {k: v.cpu() for k, v in S_.state_dict_of(bank, 'dec', E.decoder_specs(16, 2)).items()}
    return (enc, dec), hist


def _dice_of(prob, masks):
    """train.py:116-118 per image: postprocessing(threshold 0.75) then dice_coeff_2label; returns (cup, disc) means."""
    from utils.metrics import dice_coeff_2label, postprocessing
    cup = disc = 0.0
    for i in range(prob.shape[0]):
        c, d = dice_coeff_2label(postprocessing(prob[i], threshold=0.75, dataset='fundus'), masks[i])
        cup, disc = cup + c, disc + d
    n = prob.shape[0]
    return cup / n, disc / n


def _test_batches(test, bs=8):
    for b0 in range(0, len(test), bs):
        x = np.stack([t[0] for t in test[b0:b0 + bs]]).astype(np.float32).transpose(0, 3, 1, 2) / 127.5 - 1.0    # trans.Normalize
        yield torch.from_numpy(x), np.stack([t[1] for t in test[b0:b0 + bs]])


def evaluate_with_oracle(states, test):
    """Held-out domain 0, BatchNorm in EVAL mode (train.py:92-93), oracle forward on the CPU: one evaluator for all three runs."""
    from oracle import unet as OU
    enc, dec = (OU.clone_state(s) for s in states)
    cups, discs = [], []
    with torch.no_grad():
        for x, m in _test_batches(test):
            p = torch.sigmoid(OU.decoder_forward(OU.encoder_forward(x, enc, False), dec, False))
            c, d = _dice_of(p, m)
            cups.append(c * x.shape[0])
            discs.append(d * x.shape[0])
    return sum(cups) / len(test), sum(discs) / len(test)


def evaluate_with_product(states, test, device='cuda:0'):
    """The same evaluation through the drop-in modules (HIP forward, eval mode): what train.py::test_fundus runs."""
    from networks.unet import Encoder, Decoder
    enc, dec = Encoder().to(device), Decoder(num_classes=2).to(device)
    enc.load_state_dict(states[0])
    dec.load_state_dict(states[1])
    enc.eval()
    dec.eval()
    cups, discs = [], []
    with torch.no_grad():
        for x, m in _test_batches(test):
            p = torch.sigmoid(dec(enc(x.to(device)))).cpu()
            c, d = _dice_of(p, m)
            cups.append(c * x.shape[0])
            discs.append(d * x.shape[0])
    return sum(cups) / len(test), sum(discs) / len(test)
