"""oracle.unet / oracle.losses vs. fixtures produced by the reference's nn.Modules and loss functions."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import unet as OU, losses as OL
from golden_util import assert_sig_close, modules_states, state_checksum, bn_shadowed_bias, assert_noise_level

T = torch.from_numpy


def _block_state(G, tag):
    pre = tag + '.sd.'
    return {k[len(pre):]: T(G[k]).clone() for k in G.files if k.startswith(pre)}


@pytest.fixture(scope='module')
def B(golden_dir):
    return np.load(os.path.join(golden_dir, 'blocks.npz'))


def _run_block(G, tag, fn):
    sd = OU.clone_state(_block_state(G, tag), requires_grad=True)
    ins = [T(G['%s.in%d' % (tag, i)]).clone().requires_grad_(True) for i in range(2) if '%s.in%d' % (tag, i) in G.files]
    y = fn(ins, sd)
    np.testing.assert_allclose(y.detach().numpy(), G[tag + '.y'], rtol=1e-4, atol=2e-5)
    (y * T(G[tag + '.w'])).sum().backward()
    for i, t in enumerate(ins):
        ref = G['%s.din%d' % (tag, i)]
        np.testing.assert_allclose(t.grad.numpy(), ref, rtol=1e-3, atol=1e-4 * np.abs(ref).max())
    for k in OU.param_keys(sd):
        ref = G['%s.g.%s' % (tag, k)]
        g = sd[k].grad.numpy() if sd[k].grad is not None else np.zeros_like(ref)
        if 'conv' in k and k.endswith('.bias'):
            # conv bias under train-mode BN: analytically 0, the reference holds fp32 cancellation noise
            wref = np.abs(G['%s.g.%s' % (tag, k.replace('.bias', '.weight'))]).max()
            assert np.abs(g).max() < 1e-4 * (wref + 1) and np.abs(ref).max() < 1e-4 * (wref + 1), k
            continue
        np.testing.assert_allclose(g, ref, rtol=1e-3, atol=2e-4 * (np.abs(ref).max() + 0.1), err_msg=k)   # conv bias under BN: exact 0 + fp noise
    for k in sd:
        if 'running' in k or 'num_batches' in k:
            np.testing.assert_allclose(sd[k].numpy(), G['%s.after.%s' % (tag, k)], rtol=1e-5, atol=1e-6, err_msg=k)


def test_convd_blocks(B):
    _run_block(B, 'convd_first', lambda i, sd: OU.convd(i[0], _pfx(sd), 'm', True, True))
    _run_block(B, 'convd', lambda i, sd: OU.convd(i[0], _pfx(sd), 'm', False, True))
    _run_block(B, 'convd_leaky', lambda i, sd: OU.convd(i[0], _pfx(sd), 'm', False, True, slope=0.01))


class _pfx(dict):
    """View of a block-local state dict under the prefix 'm.' (oracle functions take prefixed names)."""
    def __init__(self, sd):
        super().__init__()
        self.sd = sd

    def __getitem__(self, k):
        return self.sd[k[2:]]

    def __setitem__(self, k, v):
        self.sd[k[2:]] = v

    def __contains__(self, k):
        return k[2:] in self.sd


def test_convu_blocks(B):
    _run_block(B, 'convu_first', lambda i, sd: OU.convu(i[0], i[1], _pfx(sd), 'm', True, True))
    _run_block(B, 'convu', lambda i, sd: OU.convu(i[0], i[1], _pfx(sd), 'm', False, True))


def test_convu_rec_block_dsbn_domain_selection(B):
    _run_block(B, 'convu_rec', lambda i, sd: OU.convu_rec(i[0], _pfx(sd), 'm', 2, True))
    # domains 0 and 1 untouched
    assert int(B['convu_rec.after.bn1.bns.0.num_batches_tracked']) == 0
    assert int(B['convu_rec.after.bn1.bns.2.num_batches_tracked']) == 1


def test_state_manifest_matches_reference(golden_dir):
    with open(os.path.join(golden_dir, 'state_manifest.json')) as f:
        man = json.load(f)
    for nm, sd in (('encoder', OU.encoder_state()), ('seg_decoder', OU.decoder_state()),
                   ('rec_decoder', OU.rec_decoder_state(num_classes=3, num_domains=3))):
        got = [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()]
        assert got == man[nm], nm
        assert sum(v.numel() for k, v in sd.items() if OU.is_param(k)) == man[nm + '_params']
    assert man['encoder_params'] == 1967904 and man['seg_decoder_params'] == 1217362 and man['rec_decoder_params'] == 614755


@pytest.fixture(scope='module')
def M(golden_dir):
    return np.load(os.path.join(golden_dir, 'modules.npz'))


@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_full_modules_forward_backward(M, mode):
    enc, dec, rec, chks = modules_states()
    for c, nm in zip(chks, ('enc', 'dec', 'rec')):
        np.testing.assert_allclose(c, M['chk.' + nm], rtol=1e-12)        # same generated weights as the fixture
    training = mode == 'train'
    enc, dec, rec = (OU.clone_state(s, requires_grad=True) for s in (enc, dec, rec))
    x = T(M['x']).clone().requires_grad_(True)
    feats = OU.encoder_forward(x, enc, training)
    logits = OU.decoder_forward(feats, dec, training)
    r1 = OU.rec_decoder_forward(feats[-1][0:2], rec, 1, training)
    r2 = OU.rec_decoder_forward(feats[-1][2:4], rec, 2, training)
    for i, f in enumerate(feats):
        ref = M['%s.feat%d' % (mode, i + 1)]
        np.testing.assert_allclose(f.detach().numpy(), ref, rtol=1e-3, atol=1e-4 * np.abs(ref).max())
    for got, key in ((logits, 'logits'), (r1, 'rec_d1'), (r2, 'rec_d2')):
        ref = M['%s.%s' % (mode, key)]
        np.testing.assert_allclose(got.detach().numpy(), ref, rtol=1e-3, atol=2e-4 * np.abs(ref).max())
    if not training:
        return
    loss = (logits * T(M['wl'])).sum() + (r1 * T(M['wr0'])).sum() + (r2 * T(M['wr1'])).sum()
    loss.backward()
    ref = M['train.dx']
    np.testing.assert_allclose(x.grad.numpy(), ref, rtol=1e-2, atol=1e-3 * np.abs(ref).max())
    for nm, sd in (('enc', enc), ('dec', dec), ('rec', rec)):
        for k in OU.param_keys(sd):
            g = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
            if bn_shadowed_bias(k):
                assert_noise_level(g, M['train.g%s.sig.%s' % (nm, k.replace('.bias', '.weight'))], k)
                continue
            assert_sig_close(g, M['train.g%s.sig.%s' % (nm, k)], 2e-3, name=nm + '.' + k)
            fk = 'train.g%s.full.%s' % (nm, k)
            if fk in M.files:
                np.testing.assert_allclose(g.numpy(), M[fk], rtol=1e-2, atol=2e-3 * (np.abs(M[fk]).max() + 1e-6), err_msg=fk)
        for k in sd:
            if 'running' in k or 'num_batches' in k:
                np.testing.assert_allclose(sd[k].numpy(), M['train.buf.%s.%s' % (nm, k)], rtol=1e-4, atol=1e-6, err_msg=k)


def test_losses(golden_dir):
    G = np.load(os.path.join(golden_dir, 'losses.npz'))
    l1 = T(G['logit1']).clone().requires_grad_(True)
    mask = T(G['mask'])
    p1 = torch.sigmoid(l1)
    for nm, v in (('bce1', OL.bce(p1, mask)), ('dice1', OL.dice_loss(p1, mask))):
        np.testing.assert_allclose(v.item(), float(G[nm]), rtol=1e-6)
        g = torch.autograd.grad(v, [l1], retain_graph=True)[0]
        np.testing.assert_allclose(g.numpy(), G[nm + '.g0'], rtol=1e-4, atol=1e-9)
    c1 = T(G['c1']).clone().requires_grad_(True)
    c2 = T(G['c2']).clone().requires_grad_(True)
    q1, q2 = torch.sigmoid(c1), torch.sigmoid(c2)
    v = OL.kd(q2, q1)
    np.testing.assert_allclose(v.item(), float(G['kd']), rtol=1e-5)
    # closed form used by the fused HIP loss kernel
    np.testing.assert_allclose(((q1 - q2) * (q1.log() - q2.log())).mean().item(), float(G['kd']), rtol=1e-5)
    g = torch.autograd.grad(v, [c1, c2])
    np.testing.assert_allclose(g[0].numpy(), G['kd.g0'], rtol=1e-4, atol=1e-9)
    np.testing.assert_allclose(g[1].numpy(), G['kd.g1'], rtol=1e-4, atol=1e-9)
    lg = T(G['p.logit']).clone().requires_grad_(True)
    tgt = T(G['p.target'])
    v = OL.dice_loss_multi(torch.softmax(lg, 1), tgt, 2, ignore_index=0)
    np.testing.assert_allclose(v.item(), float(G['p.dice_multi']), rtol=1e-6)
    np.testing.assert_allclose(torch.autograd.grad(v, [lg])[0].numpy(), G['p.dice_multi.g'], rtol=1e-4, atol=1e-9)
