"""oracle.step -- torch-CPU restatement of one RAM-DSIR training step.  TEST INFRASTRUCTURE ONLY.

Restates, from the reference checkout:
  train_fundus  step body   code/train.py:225-296   (sigmoid / BCELoss / dice_loss)
  train_prostate step body  code/train.py:393-465   (softmax / CrossEntropyLoss / dice_loss_multi(ignore_index=0))
  optimizer                 code/train.py:573-576   Adam(betas=(0.9,0.999)), 3 groups, encoder at lr/2
  poly LR                   code/train.py:289-293   lr*(1-iter/total)^0.9, written AFTER optimizer.step()

Only the flag combination that runs end-to-end in the reference (``--ram --rec``, optional
``--consistency``) is restated.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import losses as L
from . import unet as U


class StepConfig:
    def __init__(self, dataset='fundus', batch_sizes=(2, 3, 3), lambda_rec=0.1, consistency='kd',
                 lr=2e-3, total_iters=21200, num_classes=2, slope=0.0):
        self.dataset = dataset
        self.batch_sizes = list(batch_sizes)
        self.lambda_rec = lambda_rec
        self.consistency = consistency          # 'kd' | 'mse' | None (= no --consistency)
        self.lr = lr
        self.total_iters = total_iters
        self.num_classes = num_classes
        self.slope = slope


def poly_lr(cfg, iter_num):
    """LR in force for the optimizer.step() of iteration ``iter_num`` (train.py:289: the value is
    computed from the PREVIOUS iteration's iter_num and written after that step)."""
    if iter_num == 0:
        return cfg.lr
    return cfg.lr * (1 - (iter_num - 1) / cfg.total_iters) ** 0.9


def forward_losses(enc, dec, rec, img, img_freq, mask, cfg):
    """Returns (total loss, dict of components, dict of intermediates).  train.py:246-283 / 412-451."""
    out = OrderedDict()
    feats1 = U.encoder_forward(img, enc, True, cfg.slope)
    logit1 = U.decoder_forward(feats1, dec, True, cfg.slope)
    feats2 = U.encoder_forward(img_freq, enc, True, cfg.slope)
    logit2 = U.decoder_forward(feats2, dec, True, cfg.slope)
    if cfg.dataset == 'fundus':
        p1, p2 = torch.sigmoid(logit1), torch.sigmoid(logit2)
        seg1, seg2 = L.bce(p1, mask), L.bce(p2, mask)
        d1, d2 = L.dice_loss(p1, mask), L.dice_loss(p2, mask)
    else:
        p1, p2 = torch.softmax(logit1, 1), torch.softmax(logit2, 1)
        seg1, seg2 = F.cross_entropy(logit1, mask), F.cross_entropy(logit2, mask)
        d1 = L.dice_loss_multi(p1, mask, cfg.num_classes, ignore_index=0)
        d2 = L.dice_loss_multi(p2, mask, cfg.num_classes, ignore_index=0)
    if cfg.consistency == 'kd':
        cons = L.kd(p2, p1)
    elif cfg.consistency == 'mse':
        cons = F.mse_loss(p2, p1)
    else:
        cons = torch.zeros(())
    loss = 0
    rec_losses, rec_soft = [], []
    left = 0
    for d, b in enumerate(cfg.batch_sizes):
        right = left + b
        r = torch.tanh(U.rec_decoder_forward(feats2[-1][left:right], rec, d, True, cfg.slope))
        lr_ = F.mse_loss(r, U.q(img[left:right]))                  # target is the ORIGINAL image (train.py:273); U.q: identity unless the bf16 rounding model is on
        loss = loss + cfg.lambda_rec * lr_
        rec_losses.append(lr_)
        rec_soft.append(r)
        left = right
    loss = loss + seg1 + seg2 + d1 + d2 + 0.5 * cons               # train.py:283
    comps = OrderedDict(seg1=seg1, dice1=d1, seg2=seg2, dice2=d2, cons=cons,
                        rec=torch.stack(rec_losses), total=loss)
    inter = OrderedDict(logit1=logit1, logit2=logit2, feats1=feats1, feats2=feats2,
                        rec_soft=torch.cat(rec_soft, 0))
    return loss, comps, inter


def adam_state(params):
    return OrderedDict((k, dict(m=torch.zeros_like(p), v=torch.zeros_like(p))) for k, p in params.items())


def adam_update(p, g, st, lr, t, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor math (no amsgrad, no weight decay); t is 1-based."""
    st['m'].mul_(beta1).add_(g, alpha=1 - beta1)
    st['v'].mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** t
    bc2 = 1 - beta2 ** t
    denom = (st['v'].sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(st['m'], denom, value=-(lr / bc1))


def train_step(enc, dec, rec, opt, img, img_freq, mask, cfg, iter_num):
    """One full step, in place on the three state dicts and the Adam state ``opt`` =
    {'enc': adam_state, 'dec': ..., 'rec': ...}.  Returns (components, grads)."""
    groups = (('enc', enc, 0.5), ('dec', dec, 1.0), ('rec', rec, 1.0))
    leaves = {}
    for gname, sd, _ in groups:
        for k in U.param_keys(sd):
            sd[k] = sd[k].detach().requires_grad_(True)
            leaves[(gname, k)] = sd[k]
    loss, comps, _ = forward_losses(enc, dec, rec, img, img_freq, mask, cfg)
    keys = list(leaves)
    grads = torch.autograd.grad(loss, [leaves[k] for k in keys], allow_unused=True)
    lr = poly_lr(cfg, iter_num)
    gout = {}
    with torch.no_grad():
        for (gname, k), g in zip(keys, grads):
            sd = dict(enc=enc, dec=dec, rec=rec)[gname]
            p = sd[k].detach()
            sd[k] = p
            if g is None:
                continue
            gout[(gname, k)] = g
            mult = 0.5 if gname == 'enc' else 1.0
            adam_update(p, g, opt[gname][k], lr * mult, iter_num + 1)
    return OrderedDict((k, v.detach()) for k, v in comps.items()), gout
