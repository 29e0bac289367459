#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by IMPORTING THE REFERENCE (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--ref /root/reference]

The reference's Python never travels to the GPU box: only the data written here does.  Each fixture
is inputs + the outputs the reference's own code produced for them.  Model weights are not stored:
they are regenerated from ``oracle.unet.*_state(seed)`` (torch CPU generator) and loaded into the
reference's nn.Modules with ``load_state_dict`` -- a checksum of every state is stored so a drift in
the generator is detected instead of silently comparing different networks.

What runs from the reference:  networks.unet.{Encoder,Decoder,Rec_Decoder} (+ networks.dsbn),
utils.losses.{dice_loss,dice_loss_multi}, dataset.fundus.{extract_amp_spectrum,low_freq_mutate_np,
source_to_target_freq}, dataset.transform.to_multilabel.  train.py cannot be imported here
(tensorboardX / torchvision / medpy / SimpleITK are absent), so its step body (train.py:225-296,
393-465) is driven below with torch's own BCELoss / CrossEntropyLoss / KLDivLoss / MSELoss / Adam
exactly as train.py wires them.
"""
import argparse
import json
import os
import random
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import unet as OU          # noqa: E402  (only for the seeded state generator)


def sig(t):
    """Signature of a tensor: (sum, abs-sum, sum of squares, numel) in float64 + first 8 elements."""
    a = t.detach().double().reshape(-1)
    return np.array([a.sum().item(), a.abs().sum().item(), (a * a).sum().item(), a.numel()]
                    + a[:8].tolist() + [0.0] * max(0, 8 - a.numel()), dtype=np.float64)


def state_checksum(sd):
    return np.array([sum(v.double().sum().item() for v in sd.values()),
                     sum(v.double().abs().sum().item() for v in sd.values())])


def pack_grads(prefix, named_grads, out, full_limit=4096):
    for k, g in named_grads:
        out['%s.sig.%s' % (prefix, k)] = sig(g)
        if g.numel() <= full_limit:
            out['%s.full.%s' % (prefix, k)] = g.detach().numpy().copy()


# ------------------------------------------------------------------------------------------ RAM
def gen_ram(ref_fundus):
    out = {}
    cases = []
    rng = np.random.RandomState(1337)
    shapes = [(16, 16), (20, 20), (24, 24), (30, 30), (40, 40), (48, 48), (50, 50), (24, 32), (64, 64)]
    lam_seeds = {}
    # find python-random seeds that give each lambda in {0.1,...,1.0}
    for s in range(200):
        random.seed(s)
        lam = random.randint(1, 10) / 10
        lam_seeds.setdefault(lam, s)
    for ci, (h, w) in enumerate(shapes):
        for lam in (0.1, 0.5, 1.0) if ci < 4 else (0.3, 0.8):
            src = rng.uniform(0, 255, size=(h, w, 3)).astype(np.float32)
            trg = rng.uniform(0, 255, size=(h, w, 3)).astype(np.float32)
            if ci == 1 and lam == 0.5:
                src[..., 1] = 0.0                      # zero-amplitude source channel: angle()==0 branch
            if ci == 2 and lam == 0.1:
                src = np.round(src)                    # uint8-valued pixels, as decoded PNGs are
                trg = np.round(trg)
            name = 'c%d_l%02d' % (ci, int(lam * 10))
            res = {}
            for tag, dt in (('f64', np.float64), ('f32', np.float32)):
                random.seed(lam_seeds[lam])
                amp_trg = ref_fundus.extract_amp_spectrum(trg.astype(dt).transpose(2, 0, 1))
                freq = ref_fundus.source_to_target_freq(src.astype(dt), amp_trg, L=0.1)
                res[tag] = (amp_trg, freq)
            out[name + '.src'] = src
            out[name + '.trg'] = trg
            out[name + '.lam'] = np.array(lam)
            out[name + '.amp_trg'] = res['f64'][0]
            out[name + '.freq_f64'] = res['f64'][1]
            out[name + '.freq_f32'] = res['f32'][1].astype(np.float32)
            # low_freq_mutate_np alone (amplitudes in, amplitudes out)
            random.seed(lam_seeds[lam])
            a_s = np.abs(np.fft.fft2(src.astype(np.float64).transpose(2, 0, 1), axes=(-2, -1)))
            out[name + '.mutated'] = ref_fundus.low_freq_mutate_np(a_s.copy(), res['f64'][0].copy(), L=0.1)
            cases.append(name)
    out['cases'] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, 'ram.npz'), **out)
    print('ram.npz', len(cases), 'cases')


# ------------------------------------------------------------------------------------------ modules
def load_ref(mod, sd):
    missing = mod.load_state_dict(sd, strict=True)
    return mod


def gen_modules(RU):
    out = {}
    torch.manual_seed(7)
    B, S = 4, 32
    enc_sd, dec_sd, rec_sd = OU.encoder_state(seed=10), OU.decoder_state(seed=11), OU.rec_decoder_state(seed=12)
    out['chk.enc'], out['chk.dec'], out['chk.rec'] = map(state_checksum, (enc_sd, dec_sd, rec_sd))
    # non-trivial BN affine so that gamma/beta paths are exercised
    g = torch.Generator().manual_seed(99)
    for sd in (enc_sd, dec_sd, rec_sd):
        for k in sd:
            if ('.bn' in k) and k.endswith('.weight'):
                sd[k] = 1.0 + 0.2 * torch.randn(sd[k].shape, generator=g)
            if ('.bn' in k) and k.endswith('.bias'):
                sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    x = torch.randn(B, 3, S, S, generator=g)
    out['x'] = x.numpy()
    for mode in ('train', 'eval'):
        enc = load_ref(RU.Encoder(), OU.clone_state(enc_sd))
        dec = load_ref(RU.Decoder(num_classes=2), OU.clone_state(dec_sd))
        rec = load_ref(RU.Rec_Decoder(num_classes=3, norm='dsbn', num_domains=3), OU.clone_state(rec_sd))
        for m in (enc, dec, rec):
            m.train() if mode == 'train' else m.eval()
        xin = x.clone().requires_grad_(True)
        feats = enc(xin)
        logits = dec(feats)
        recs = [rec(feats[-1][0:2], domain_label=1 * torch.ones(2, dtype=torch.long)),
                rec(feats[-1][2:4], domain_label=2 * torch.ones(2, dtype=torch.long))]
        for i, f in enumerate(feats):
            out['%s.feat%d' % (mode, i + 1)] = f.detach().numpy()
        out['%s.logits' % mode] = logits.detach().numpy()
        out['%s.rec_d1' % mode] = recs[0].detach().numpy()
        out['%s.rec_d2' % mode] = recs[1].detach().numpy()
        if mode == 'train':
            wl = torch.randn(logits.shape, generator=g)
            wr = [torch.randn(r.shape, generator=g) for r in recs]
            out['wl'], out['wr0'], out['wr1'] = wl.numpy(), wr[0].numpy(), wr[1].numpy()
            loss = (logits * wl).sum() + (recs[0] * wr[0]).sum() + (recs[1] * wr[1]).sum()
            loss.backward()
            out['train.dx'] = xin.grad.numpy()
            pack_grads('train.genc', [(k, p.grad) for k, p in enc.named_parameters()], out)
            pack_grads('train.gdec', [(k, p.grad) for k, p in dec.named_parameters()], out)
            pack_grads('train.grec', [(k, p.grad if p.grad is not None else torch.zeros_like(p))
                                      for k, p in rec.named_parameters()], out)
            for nm, m in (('enc', enc), ('dec', dec), ('rec', rec)):
                for k, v in m.state_dict().items():
                    if 'running' in k or 'num_batches' in k:
                        out['train.buf.%s.%s' % (nm, k)] = v.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'modules.npz'), **out)
    print('modules.npz', len(out), 'arrays')

    manifest = {}
    for nm, m in (('encoder', RU.Encoder()), ('seg_decoder', RU.Decoder(num_classes=2)),
                  ('rec_decoder', RU.Rec_Decoder(num_classes=3, norm='dsbn', num_domains=3))):
        manifest[nm] = [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]
        manifest[nm + '_params'] = sum(p.numel() for p in m.parameters())
    with open(os.path.join(HERE, 'state_manifest.json'), 'w') as f:
        json.dump(manifest, f, indent=0)
    print('state_manifest.json')


def gen_modules_norm(RU):
    """normalization(planes, 'gn' / 'in') (unet.py:20-23): the reference's Encoder + Decoder built with --norm gn / in, train-mode
    forward and backward (nn.GroupNorm / nn.InstanceNorm2d behave the same in eval mode), non-trivial affine for gn."""
    out = {}
    B, S = 3, 32
    g = torch.Generator().manual_seed(123)
    x = torch.randn(B, 3, S, S, generator=g)
    out['x'] = x.numpy()
    for norm in ('gn', 'in'):
        enc_sd, dec_sd = OU.encoder_state(n=8, seed=20, norm=norm), OU.decoder_state(n=8, seed=21, norm=norm)
        for sd in (enc_sd, dec_sd):
            for k in sd:
                if ('.bn' in k) and k.endswith('.weight'):
                    sd[k] = 1.0 + 0.2 * torch.randn(sd[k].shape, generator=g)
                if ('.bn' in k) and k.endswith('.bias'):
                    sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
        out['%s.chk.enc' % norm], out['%s.chk.dec' % norm] = state_checksum(enc_sd), state_checksum(dec_sd)
        for nm, sd in (('enc', enc_sd), ('dec', dec_sd)):
            for k, v in sd.items():
                if '.bn' in k:
                    out['%s.sd.%s.%s' % (norm, nm, k)] = v.numpy().copy()
        enc = load_ref(RU.Encoder(n=8, norm=norm), OU.clone_state(enc_sd))
        dec = load_ref(RU.Decoder(n=8, num_classes=2, norm=norm), OU.clone_state(dec_sd))
        enc.train(); dec.train()
        xin = x.clone().requires_grad_(True)
        feats = enc(xin)
        logits = dec(feats)
        for i, f in enumerate(feats):
            out['%s.feat%d' % (norm, i + 1)] = f.detach().numpy()
        out['%s.logits' % norm] = logits.detach().numpy()
        wl = torch.randn(logits.shape, generator=g)
        wf = torch.randn(feats[2].shape, generator=g)                     # a second consumer of an encoder feature
        out['%s.wl' % norm], out['%s.wf' % norm] = wl.numpy(), wf.numpy()
        ((logits * wl).sum() + (feats[2] * wf).sum()).backward()
        out['%s.dx' % norm] = xin.grad.numpy()
        pack_grads('%s.genc' % norm, [(k, p.grad) for k, p in enc.named_parameters()], out, full_limit=1 << 14)
        pack_grads('%s.gdec' % norm, [(k, p.grad) for k, p in dec.named_parameters()], out, full_limit=1 << 14)
        out['%s.keys.enc' % norm] = np.array(list(enc.state_dict().keys()))
        out['%s.keys.dec' % norm] = np.array(list(dec.state_dict().keys()))
    np.savez_compressed(os.path.join(HERE, 'modules_norm.npz'), **out)
    print('modules_norm.npz', len(out), 'arrays')


# ------------------------------------------------------------------------------------------ single blocks
def gen_blocks(RU):
    """ConvD / ConvU / ConvU_Rec forward + backward on tiny shapes (module-local goldens, weights stored)."""
    out = {}
    g = torch.Generator().manual_seed(5)

    def rnd_bn(m):
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.data = 1.0 + 0.3 * torch.randn(mod.weight.shape, generator=g)
                mod.bias.data = 0.2 * torch.randn(mod.bias.shape, generator=g)

    def run(tag, m, inputs, kwargs=None):
        rnd_bn(m)
        for k, v in m.state_dict().items():
            out['%s.sd.%s' % (tag, k)] = v.numpy().copy()
        m.train()
        ins = [t.clone().requires_grad_(True) for t in inputs]
        y = m(*ins, **(kwargs or {}))
        w = torch.randn(y.shape, generator=g)
        (y * w).sum().backward()
        out[tag + '.y'] = y.detach().numpy()
        out[tag + '.w'] = w.numpy()
        for i, t in enumerate(ins):
            out['%s.in%d' % (tag, i)] = inputs[i].numpy()
            out['%s.din%d' % (tag, i)] = t.grad.numpy()
        for k, p in m.named_parameters():
            out['%s.g.%s' % (tag, k)] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
        for k, v in m.state_dict().items():
            if 'running' in k or 'num_batches' in k:
                out['%s.after.%s' % (tag, k)] = v.numpy().copy()

    torch.manual_seed(3)
    run('convd_first', RU.ConvD(3, 16, 'bn', first=True), [torch.randn(3, 3, 16, 16, generator=g)])
    run('convd', RU.ConvD(16, 32, 'bn'), [torch.randn(3, 16, 16, 16, generator=g)])
    run('convd_leaky', RU.ConvD(16, 32, 'bn', activation='leaky'), [torch.randn(2, 16, 16, 16, generator=g)])
    run('convu_first', RU.ConvU(64, 'bn', first=True),
        [torch.randn(2, 64, 8, 8, generator=g), torch.randn(2, 32, 16, 16, generator=g)])
    run('convu', RU.ConvU(32, 'bn'),
        [torch.randn(2, 64, 8, 8, generator=g), torch.randn(2, 16, 16, 16, generator=g)])
    run('convu_rec', RU.ConvU_Rec(64, 'dsbn', num_domains=3), [torch.randn(3, 64, 8, 8, generator=g)],
        dict(domain_label=2 * torch.ones(3, dtype=torch.long)))
    np.savez_compressed(os.path.join(HERE, 'blocks.npz'), **out)
    print('blocks.npz', len(out), 'arrays')


# ------------------------------------------------------------------------------------------ standalone DSBN
def gen_dsbn(ref_root):
    """networks/dsbn.py:24-27 on its own: forward(x, domain_label) -> (y, domain_label), train (batch statistics of
    bns[domain_label[0]], running statistics of that domain only) + backward, then eval.  Channel 2 has |mean| = 1e3 sigma."""
    import networks.dsbn as RD
    out = {}
    g = torch.Generator().manual_seed(31)
    m = RD.DomainSpecificBatchNorm2d(12, num_domains=3)
    for d, bn in enumerate(m.bns):
        bn.weight.data = 1.0 + 0.3 * torch.randn(12, generator=g)
        bn.bias.data = 0.2 * torch.randn(12, generator=g)
    for k, v in m.state_dict().items():
        out['sd.' + k] = v.numpy().copy()
    x = torch.randn(5, 12, 10, 14, generator=g)
    x[:, 2] = x[:, 2] * 1e-2 + 10.0
    lab = torch.tensor([1, 0, 2, 1, 1])                     # only lab[0] matters (dsbn.py:26)
    m.train()
    xi = x.clone().requires_grad_(True)
    y, lab_out = m(xi, lab)
    w = torch.randn(y.shape, generator=g)
    (y * w).sum().backward()
    out['x'], out['lab'], out['y'], out['w'], out['dx'] = x.numpy(), lab.numpy(), y.detach().numpy(), w.numpy(), xi.grad.numpy()
    out['lab_is_same_object'] = np.array(lab_out is lab)
    for k, p in m.named_parameters():
        out['g.' + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    for k, v in m.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            out['after.' + k] = v.numpy().copy()
    m.eval()
    y2, _ = m(x, lab)
    out['y_eval'] = y2.detach().numpy()
    np.savez_compressed(os.path.join(HERE, 'dsbn.npz'), **out)
    print('dsbn.npz', len(out), 'arrays')


# ------------------------------------------------------------------------------------------ losses
def gen_losses(ref_losses):
    from torch.nn import BCELoss, KLDivLoss, MSELoss, CrossEntropyLoss
    out = {}
    g = torch.Generator().manual_seed(21)
    B, S = 3, 12
    logit1 = (3 * torch.randn(B, 2, S, S, generator=g))
    logit2 = (3 * torch.randn(B, 2, S, S, generator=g))
    logit1[0, 0, 0, :4] = torch.tensor([60.0, -60.0, 120.0, -120.0])      # BCE log clamp / saturation
    mask = (torch.rand(B, 2, S, S, generator=g) > 0.6).float()
    out['logit1'], out['logit2'], out['mask'] = logit1.numpy(), logit2.numpy(), mask.numpy()
    l1 = logit1.clone().requires_grad_(True)
    l2 = logit2.clone().requires_grad_(True)
    p1, p2 = torch.sigmoid(l1), torch.sigmoid(l2)
    bce1 = BCELoss()(p1, mask)
    dice1 = ref_losses.dice_loss(p1, mask)
    kl = KLDivLoss()
    # saturated probabilities make log(p)=-inf in KD; evaluate KD on a clean pair
    c1 = (2 * torch.randn(B, 2, S, S, generator=g)).requires_grad_(True)
    c2 = (2 * torch.randn(B, 2, S, S, generator=g)).requires_grad_(True)
    q1, q2 = torch.sigmoid(c1), torch.sigmoid(c2)
    kdv = kl(q2.log(), q1) + kl(q1.log(), q2)                              # KD(input=q2, target=q1), train.py:259
    msev = MSELoss()(q2, q1)
    out['c1'], out['c2'] = c1.detach().numpy(), c2.detach().numpy()
    for nm, v, wrt in (('bce1', bce1, [l1]), ('dice1', dice1, [l1]), ('kd', kdv, [c1, c2]), ('mse', msev, [c1, c2])):
        gr = torch.autograd.grad(v, wrt, retain_graph=True)
        out[nm] = np.array(v.item())
        for i, gg in enumerate(gr):
            out['%s.g%d' % (nm, i)] = gg.numpy()
    # prostate flavour
    lg = (2 * torch.randn(B, 2, S, S, generator=g)).requires_grad_(True)
    tgt = (torch.rand(B, S, S, generator=g) > 0.7).long()
    ps = torch.softmax(lg, 1)
    ce = CrossEntropyLoss()(lg, tgt)
    dm = ref_losses.dice_loss_multi(ps, tgt, num_classes=2, ignore_index=0)
    out['p.logit'], out['p.target'] = lg.detach().numpy(), tgt.numpy()
    for nm, v in (('p.ce', ce), ('p.dice_multi', dm)):
        out[nm] = np.array(v.item())
        out[nm + '.g'] = torch.autograd.grad(v, [lg], retain_graph=True)[0].numpy()
    # tanh + MSE (rec)
    r = torch.randn(2, 3, S, S, generator=g).requires_grad_(True)
    t = torch.rand(2, 3, S, S, generator=g) * 2 - 1
    mv = MSELoss()(torch.tanh(r), t)
    out['r.logit'], out['r.target'], out['r.mse'] = r.detach().numpy(), t.numpy(), np.array(mv.item())
    out['r.mse.g'] = torch.autograd.grad(mv, [r])[0].numpy()
    np.savez_compressed(os.path.join(HERE, 'losses.npz'), **out)
    print('losses.npz')


# ------------------------------------------------------------------------------------------ full steps
def gen_steps(RU, ref_losses, only64=False):
    from torch.nn import BCELoss, KLDivLoss, MSELoss, CrossEntropyLoss
    from torch.optim import Adam

    def KD(input, target):                                   # train.py:85-88
        c = KLDivLoss()
        return c(input.log(), target) + c(target.log(), input)

    for dataset, bsl, S, ncls, cons in (('fundus', [2, 3, 3], 32, 2, 'kd'),
                                        ('fundus_mse', [1, 2, 1], 32, 2, 'mse'),
                                        ('prostate', [2, 2, 2, 2, 2], 32, 2, 'kd'),
                                        ('fundus64', [2, 3, 3], 64, 2, 'kd'),        # 4x4-pixel bottleneck: tighter gradient checks
                                        ('prostate96', [1, 2, 1, 1, 1], 96, 2, 'kd')):  # 5 domain groups, 6x6 bottleneck, 3 tiles per row
        if only64 and dataset not in ('fundus64', 'prostate96'):
            continue
        out = {}
        nd = len(bsl)
        B = sum(bsl)
        g = torch.Generator().manual_seed(1337)
        enc_sd, dec_sd = OU.encoder_state(seed=20), OU.decoder_state(num_classes=ncls, seed=21)
        rec_sd = OU.rec_decoder_state(num_classes=3, num_domains=nd, seed=22)
        out['chk.enc'], out['chk.dec'], out['chk.rec'] = map(state_checksum, (enc_sd, dec_sd, rec_sd))
        encoder = load_ref(RU.Encoder(), OU.clone_state(enc_sd))
        seg_decoder = load_ref(RU.Decoder(num_classes=ncls), OU.clone_state(dec_sd))
        rec_decoder = load_ref(RU.Rec_Decoder(num_classes=3, norm='dsbn', num_domains=nd), OU.clone_state(rec_sd))
        base_lr = 2e-3 if dataset.startswith('fundus') else 1e-3
        total_iters = 50
        optimizer = Adam([{"params": encoder.parameters(), 'lr': base_lr / 2},          # train.py:573-576
                          {"params": seg_decoder.parameters(), 'lr': base_lr},
                          {"params": rec_decoder.parameters(), 'lr': base_lr}],
                         lr=base_lr, betas=(0.9, 0.999))
        criterion = BCELoss() if dataset.startswith('fundus') else CrossEntropyLoss()
        rec_criterion = MSELoss()
        consistency_criterion = KD if cons == 'kd' else MSELoss()
        encoder.train(); seg_decoder.train(); rec_decoder.train()
        nsteps = 3 if S < 96 else 2
        iter_num = 0
        for it in range(nsteps):
            img_multi = torch.rand(B, 3, S, S, generator=g) * 2 - 1
            img_freq_multi = (img_multi + 0.3 * torch.randn(B, 3, S, S, generator=g)).clamp(-1, 1)
            if dataset.startswith('fundus'):
                disc = (torch.rand(B, 1, S, S, generator=g) > 0.5).float()
                cup = disc * (torch.rand(B, 1, S, S, generator=g) > 0.5).float()
                mask_multi = torch.cat([cup, disc], 1)
            else:
                mask_multi = (torch.rand(B, S, S, generator=g) > 0.7).long()
            out['s%d.img' % it], out['s%d.img_freq' % it], out['s%d.mask' % it] = \
                img_multi.numpy(), img_freq_multi.numpy(), mask_multi.numpy()
            out['s%d.lr_used' % it] = np.array([pg['lr'] for pg in optimizer.param_groups])
            # ---- train.py:246-283 (fundus) / 412-451 (prostate)
            img_feats = encoder(img_multi)
            if dataset.startswith('fundus'):
                pred_soft_1 = torch.sigmoid(seg_decoder(img_feats))
                loss_seg_1 = criterion(pred_soft_1, mask_multi)
                loss_dice_1 = ref_losses.dice_loss(pred_soft_1, mask_multi)
            else:
                pred_1 = seg_decoder(img_feats)
                pred_soft_1 = torch.softmax(pred_1, dim=1)
                loss_seg_1 = criterion(pred_1, mask_multi)
                loss_dice_1 = ref_losses.dice_loss_multi(pred_soft_1, mask_multi, num_classes=ncls, ignore_index=0)
            loss = 0
            img_freq_feats = encoder(img_freq_multi)
            if dataset.startswith('fundus'):
                pred_soft_2 = torch.sigmoid(seg_decoder(img_freq_feats))
                loss_seg_2 = criterion(pred_soft_2, mask_multi)
                loss_dice_2 = ref_losses.dice_loss(pred_soft_2, mask_multi)
            else:
                pred_2 = seg_decoder(img_freq_feats)
                pred_soft_2 = torch.softmax(pred_2, dim=1)
                loss_seg_2 = criterion(pred_2, mask_multi)
                loss_dice_2 = ref_losses.dice_loss_multi(pred_soft_2, mask_multi, num_classes=ncls, ignore_index=0)
            loss_consistency = consistency_criterion(pred_soft_2, pred_soft_1)
            left = 0
            rec_l = []
            for train_idx in range(nd):
                right = left + bsl[train_idx]
                rec_soft = torch.tanh(rec_decoder(img_freq_feats[-1][left:right, ...],
                                                  domain_label=train_idx * torch.ones(bsl[train_idx], dtype=torch.long)))
                loss_rec = rec_criterion(rec_soft, img_multi[left:right])
                loss = loss + 0.1 * loss_rec
                rec_l.append(loss_rec.item())
                left = right
            loss = loss + loss_seg_1 + loss_seg_2 + loss_dice_1 + loss_dice_2 + 0.5 * loss_consistency
            optimizer.zero_grad()
            loss.backward()
            if it == 0:
                pack_grads('s0.genc', [(k, p.grad) for k, p in encoder.named_parameters()], out)
                pack_grads('s0.gdec', [(k, p.grad) for k, p in seg_decoder.named_parameters()], out)
                pack_grads('s0.grec', [(k, p.grad) for k, p in rec_decoder.named_parameters()], out)
            optimizer.step()
            lr = base_lr * (1 - iter_num / total_iters) ** 0.9                           # train.py:289-293
            optimizer.param_groups[0]["lr"] = lr / 2
            optimizer.param_groups[1]["lr"] = lr
            optimizer.param_groups[2]["lr"] = lr
            iter_num += 1
            out['s%d.losses' % it] = np.array([loss_seg_1.item(), loss_dice_1.item(), loss_seg_2.item(),
                                               loss_dice_2.item(), loss_consistency.item(), loss.item()])
            out['s%d.rec_losses' % it] = np.array(rec_l)
            for nm, m in (('enc', encoder), ('dec', seg_decoder), ('rec', rec_decoder)):
                for k, v in m.state_dict().items():
                    out['s%d.post.%s.sig.%s' % (it, nm, k)] = sig(v)
                    if v.numel() <= 1024:
                        out['s%d.post.%s.full.%s' % (it, nm, k)] = v.numpy().copy()
        out['meta'] = np.array(json.dumps(dict(dataset=dataset, batch_sizes=bsl, S=S, num_classes=ncls,
                                               consistency=cons, base_lr=base_lr, total_iters=total_iters,
                                               nsteps=nsteps, lambda_rec=0.1)))
        np.savez_compressed(os.path.join(HERE, 'step_%s.npz' % dataset), **out)
        print('step_%s.npz' % dataset, len(out), 'arrays')


# ------------------------------------------------------------------------------------------ masks
def gen_masks(ref_transform):
    g = np.array([[0, 50, 51, 128], [200, 201, 255, 49], [100, 0, 250, 202]], dtype=np.uint8)
    __mask = g.copy()
    _mask = np.zeros([__mask.shape[0], __mask.shape[1]])                                 # fundus.py:227-239
    _mask[__mask > 200] = 255
    _mask[(__mask > 50) & (__mask < 201)] = 128
    __mask[_mask == 0] = 2
    __mask[_mask == 255] = 0
    __mask[_mask == 128] = 1
    ml = ref_transform.to_multilabel(__mask).transpose(2, 0, 1)
    np.savez_compressed(os.path.join(HERE, 'masks.npz'), gray=g, multilabel=ml.astype(np.float32))
    print('masks.npz')


# ------------------------------------------------------------------------------------------ R4 sampling
def gen_sampling(ref_fundus, ref_prostate, ref_transform):
    """The REFERENCE's Fundus_Multi / Prostate_Multi (fundus.py:160-240, prostate.py:152-202) on the synthetic trees of
    tests/synth_data.py under fixed seeds: every random draw in order (crop/scale, partner domain, partner image,
    lambda) and what __getitem__ returned (img, img_freq, mask; images subsampled 8x to keep the fixture small --
    a wrong partner or lambda changes every pixel)."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import synth_data as SD
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        base = SD.make_fundus_tree(tmp)
        tf = [ref_transform.Resize((256, 256)), ref_transform.RandomScaleCrop((256, 256))]       # train.py:541

        def compose(sample):
            for t in tf:
                sample = t(sample)
            return sample
        for tag, dom, ood, tdi in (('f_ood', [1], True, 0), ('f_ind', [2, 3], False, 1)):
            ds = ref_fundus.Fundus_Multi(domain_idx_list=dom, base_dir=base, split='train', transform=compose,
                                         is_out_domain=ood, test_domain_idx=tdi)
            random.seed(1337)
            np.random.seed(1337)
            n = min(len(ds), 6)
            for i in range(n):
                with SD.DrawLog() as dl:
                    img, frq, mask = ds[i]
                out['%s.%d.draws' % (tag, i)] = np.array(dl.log)
                out['%s.%d.img' % (tag, i)] = img.numpy()[:, ::8, ::8].copy()
                out['%s.%d.frq' % (tag, i)] = frq.numpy()[:, ::8, ::8].copy()
                out['%s.%d.mask' % (tag, i)] = mask.numpy()[:, ::4, ::4].astype(np.uint8)
                out['%s.%d.sig' % (tag, i)] = np.stack([sig(img), sig(frq), sig(mask)])
            out[tag + '.n'] = np.array(n)
        pbase = SD.make_prostate_tree(tmp)
        real_listdir = os.listdir

        def logged_listdir(p):
            r = sorted(real_listdir(p))                       # fixed order: the file system's own order does not travel
            return r
        os.listdir = logged_listdir
        try:
            ds = ref_prostate.Prostate_Multi(domain_idx_list=[0, 2], base_dir=pbase, split='train', is_out_domain=True,
                                             test_domain_idx=4)
            random.seed(1337)
            np.random.seed(1337)
            n = min(len(ds), 6)
            for i in range(n):
                with SD.DrawLog() as dl:
                    img, frq, mask = ds[i]
                out['p_ood.%d.draws' % i] = np.array(dl.log)
                out['p_ood.%d.img' % i] = img.numpy()[:, ::4, ::4].copy()
                out['p_ood.%d.frq' % i] = frq.numpy()[:, ::4, ::4].copy()
                out['p_ood.%d.sig' % i] = np.stack([sig(img), sig(frq), sig(mask.float())])
            out['p_ood.n'] = np.array(n)
        finally:
            os.listdir = real_listdir
    np.savez_compressed(os.path.join(HERE, 'sampling.npz'), **out)
    print('sampling.npz', len(out), 'arrays')


# ------------------------------------------------------------------------------------------ evaluation metrics
def gen_metrics(ref_root):
    """utils/metrics.py:55-109 (dice_coefficient_numpy, dice_coeff_2label: the Fundus Dice of BASELINE.json) and
    :33-53 (dice, dice_multi) on synthetic masks, incl. empty / full / disjoint cases.  The module does
    ``from medpy import metric`` at import and only calculate_metric_percase touches it: an EMPTY placeholder module
    satisfies the import (nothing from medpy is executed or emulated)."""
    import types
    ph = types.ModuleType('medpy')
    ph.metric = types.ModuleType('medpy.metric')
    sys.modules.setdefault('medpy', ph)
    sys.modules.setdefault('medpy.metric', ph.metric)
    import importlib.util
    spec = importlib.util.spec_from_file_location('ref_metrics', os.path.join(ref_root, 'code', 'utils', 'metrics.py'))
    RM = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(RM)
    rng = np.random.RandomState(5)
    out = {}
    H = W = 24
    yy, xx = np.mgrid[0:H, 0:W]
    cases = []
    for i in range(10):
        if i == 0:
            p, t = np.zeros((H, W), bool), np.zeros((H, W), bool)             # both empty -> (0+1)/(1+0+0) = 1
        elif i == 1:
            p, t = np.ones((H, W), bool), np.ones((H, W), bool)
        elif i == 2:
            p, t = np.zeros((H, W), bool), np.ones((H, W), bool)
        elif i == 3:
            p, t = xx < 5, xx > 15                                             # disjoint
        else:
            c = rng.uniform(6, 18, 4)
            r = rng.uniform(3, 9, 2)
            p = (yy - c[0]) ** 2 + (xx - c[1]) ** 2 < r[0] ** 2
            t = (yy - c[2]) ** 2 + (xx - c[3]) ** 2 < r[1] ** 2
        out['d%d.pred' % i], out['d%d.gt' % i] = p.astype(np.uint8), t.astype(np.uint8)
        out['d%d.dice' % i] = np.array(RM.dice_coefficient_numpy(p, t))
        cases.append(i)
    out['ncases'] = np.array(len(cases))
    # dice_coeff_2label: (2,H,W) single sample and (B,2,H,W) batch means
    pred = (rng.uniform(size=(5, 2, H, W)) > 0.6)
    targ = (rng.uniform(size=(5, 2, H, W)) > 0.5)
    pred[1] = False
    targ[2] = False
    out['b.pred'], out['b.gt'] = pred.astype(np.uint8), targ.astype(np.uint8)
    out['b.single'] = np.array(RM.dice_coeff_2label(pred[0].astype(np.float32), torch.from_numpy(targ[0].astype(np.float32))))
    out['b.batch'] = np.array(RM.dice_coeff_2label(pred.astype(np.float32), torch.from_numpy(targ.astype(np.float32))))
    # torch dice / dice_multi (metrics.py:28-53)
    a = torch.from_numpy((rng.uniform(size=(3, H, W)) > 0.5).astype(np.float32))
    b = torch.from_numpy((rng.uniform(size=(3, H, W)) > 0.5).astype(np.float32))
    out['t.a'], out['t.b'] = a.numpy(), b.numpy()
    out['t.dice'] = np.array(RM.dice(a, b).item())
    li = torch.from_numpy(rng.randint(0, 3, (2, H, W)))
    lt = torch.from_numpy(rng.randint(0, 3, (2, H, W)))
    out['m.a'], out['m.b'] = li.numpy(), lt.numpy()
    out['m.dice_multi'] = np.array(float(RM.dice_multi(li, lt, num_classes=3)))
    out['m.dice_multi_ign0'] = np.array(float(RM.dice_multi(li, lt, num_classes=3, ignore_index=0)))
    np.savez_compressed(os.path.join(HERE, 'metrics.npz'), **out)
    print('metrics.npz', len(out), 'arrays')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--only', default='', help='comma list of fixture groups (ram,masks,losses,blocks,modules,modules_norm,steps,steps64,sampling,metrics,dsbn)')
    args = ap.parse_args()
    only = set(filter(None, args.only.split(',')))
    sys.path.insert(0, os.path.join(args.ref, 'code'))
    import networks.unet as RU
    import utils.losses as ref_losses
    import dataset.fundus as ref_fundus
    import dataset.prostate as ref_prostate
    import dataset.transform as ref_transform
    torch.set_num_threads(4)
    want = lambda g: not only or g in only
    if want('ram'):
        gen_ram(ref_fundus)
    if want('masks'):
        gen_masks(ref_transform)
    if want('losses'):
        gen_losses(ref_losses)
    if want('blocks'):
        gen_blocks(RU)
    if want('modules'):
        gen_modules(RU)
    if want('modules_norm'):
        gen_modules_norm(RU)
    if want('steps') or want('steps64'):
        gen_steps(RU, ref_losses, only64=not want('steps'))
    if want('sampling'):
        gen_sampling(ref_fundus, ref_prostate, ref_transform)
    if want('metrics'):
        gen_metrics(args.ref)
    if want('dsbn'):
        gen_dsbn(args.ref)
    with open(os.path.join(HERE, 'VERSIONS.json'), 'w') as f:
        json.dump(dict(torch=torch.__version__, numpy=np.__version__, python=sys.version.split()[0]), f)


if __name__ == '__main__':
    main()
