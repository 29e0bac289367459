"""-m gpu: "Level 1" of INTEGRATION.md -- the reference's own training loop, unchanged in structure, over the DROP-IN
modules: torch.optim.Adam with the three parameter groups of train.py:573-576, torch's BCELoss / CrossEntropyLoss /
KLDivLoss / MSELoss and utils.losses.dice_loss exactly as train.py:196-203,246-296 wires them, single-device
nn.DataParallel wrappers (train.py:205-208), poly LR written after optimizer.step().  Only Encoder / Decoder /
Rec_Decoder forward + backward are HIP (fp32 storage); everything else is PyTorch, as it would be for a user who swaps
`networks/` only.  Three consecutive steps against tests/golden/step_fundus.npz / step_prostate.npz (produced by the
reference modules under the same loop)."""
import numpy as np
import pytest
import torch
from torch.nn import BCELoss, CrossEntropyLoss, KLDivLoss, MSELoss
from torch.optim import Adam

pytestmark = pytest.mark.gpu

from golden_util import load_step, step_states, bn_shadowed_bias          # noqa: E402
from networks.unet import Encoder, Decoder, Rec_Decoder                   # noqa: E402
from utils.losses import dice_loss, dice_loss_multi                       # noqa: E402

T = torch.from_numpy


def KD(input, target):                                                    # train.py:85-88
    c = KLDivLoss()
    return c(input.log(), target) + c(target.log(), input)


@pytest.mark.parametrize('name', ['fundus', 'prostate'])
def test_reference_style_loop_over_the_drop_in_modules(golden_dir, name):
    G, meta = load_step(golden_dir, name)
    enc_sd, dec_sd, rec_sd = step_states(meta)
    bsl, ncls, nd = meta['batch_sizes'], meta['num_classes'], len(meta['batch_sizes'])
    fundus = name.startswith('fundus')
    encoder, seg_decoder = Encoder(), Decoder(num_classes=ncls)
    rec_decoder = Rec_Decoder(num_classes=3, norm='dsbn', num_domains=nd)
    for m, sd in ((encoder, enc_sd), (seg_decoder, dec_sd), (rec_decoder, rec_sd)):
        m.load_state_dict(sd, strict=True)
    encoder = torch.nn.DataParallel(encoder, device_ids=[0]).cuda()                         # train.py:205-208
    seg_decoder = torch.nn.DataParallel(seg_decoder, device_ids=[0]).cuda()
    rec_decoder = torch.nn.DataParallel(rec_decoder, device_ids=[0]).cuda()
    base_lr, total_iters = meta['base_lr'], meta['total_iters']
    optimizer = Adam([{"params": encoder.parameters(), 'lr': base_lr / 2},                  # train.py:573-576
                      {"params": seg_decoder.parameters(), 'lr': base_lr},
                      {"params": rec_decoder.parameters(), 'lr': base_lr}], lr=base_lr, betas=(0.9, 0.999))
    criterion = BCELoss() if fundus else CrossEntropyLoss()
    rec_criterion = MSELoss()
    encoder.train(); seg_decoder.train(); rec_decoder.train()
    iter_num = 0
    for it in range(meta['nsteps']):
        img_multi, img_freq_multi = T(G['s%d.img' % it]).cuda(), T(G['s%d.img_freq' % it]).cuda()
        mask_multi = T(G['s%d.mask' % it]).cuda()
        np.testing.assert_allclose([pg['lr'] for pg in optimizer.param_groups], G['s%d.lr_used' % it], rtol=1e-12)
        img_feats = encoder(img_multi)
        if fundus:
            pred_soft_1 = torch.sigmoid(seg_decoder(img_feats))
            loss_seg_1, loss_dice_1 = criterion(pred_soft_1, mask_multi), dice_loss(pred_soft_1, mask_multi)
        else:
            pred_1 = seg_decoder(img_feats)
            pred_soft_1 = torch.softmax(pred_1, dim=1)
            loss_seg_1 = criterion(pred_1, mask_multi)
            loss_dice_1 = dice_loss_multi(pred_soft_1, mask_multi, num_classes=ncls, ignore_index=0)
        loss = 0
        img_freq_feats = encoder(img_freq_multi)
        if fundus:
            pred_soft_2 = torch.sigmoid(seg_decoder(img_freq_feats))
            loss_seg_2, loss_dice_2 = criterion(pred_soft_2, mask_multi), dice_loss(pred_soft_2, mask_multi)
        else:
            pred_2 = seg_decoder(img_freq_feats)
            pred_soft_2 = torch.softmax(pred_2, dim=1)
            loss_seg_2 = criterion(pred_2, mask_multi)
            loss_dice_2 = dice_loss_multi(pred_soft_2, mask_multi, num_classes=ncls, ignore_index=0)
        loss_consistency = KD(pred_soft_2, pred_soft_1)
        left, rec_l = 0, []
        for train_idx in range(nd):                                                          # train.py:265-276
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
        got = [loss_seg_1.item(), loss_dice_1.item(), loss_seg_2.item(), loss_dice_2.item(), loss_consistency.item(), loss.item()]
        # steps >= 1: Adam's sign-like first update amplifies fp32 noise in near-zero gradients (tests/test_oracle_step.py)
        np.testing.assert_allclose(got, G['s%d.losses' % it], rtol=1e-4 if it == 0 else 5e-2)
        np.testing.assert_allclose(rec_l, G['s%d.rec_losses' % it], rtol=1e-4 if it == 0 else 5e-2)
        if it == 0:
            for tag, mod in (('enc', encoder), ('dec', seg_decoder), ('rec', rec_decoder)):
                for k, p in mod.module.named_parameters():
                    ref = G['s0.g%s.sig.%s' % (tag, k)]
                    g = p.grad.double() if p.grad is not None else torch.zeros_like(p).double()
                    if bn_shadowed_bias(k):
                        assert float(g.abs().max()) == 0.0, k
                        continue
                    np.testing.assert_allclose(float(g.norm()), np.sqrt(ref[2]), rtol=4e-2, err_msg=k)
                    fk = 's0.g%s.full.%s' % (tag, k)
                    if fk in G.files:
                        r = T(G[fk]).double()
                        assert float((g.cpu() - r).norm() / (r.norm() + 1e-30)) <= 4e-2, k
        optimizer.step()
        lr = base_lr * (1 - iter_num / total_iters) ** 0.9                                   # train.py:289-293
        optimizer.param_groups[0]["lr"] = lr / 2
        optimizer.param_groups[1]["lr"] = lr
        optimizer.param_groups[2]["lr"] = lr
        iter_num += 1
    # running statistics after three steps: plain fp32 momentum averages of per-batch statistics
    for tag, mod in (('enc', encoder), ('dec', seg_decoder), ('rec', rec_decoder)):
        sd = mod.module.state_dict()                                                         # train.py:343: unwrapped state_dicts
        for k, v in sd.items():
            if 'num_batches_tracked' in k:
                ref = G['s2.post.%s.sig.%s' % (tag, k)]
                assert float(v) == ref[4], k
