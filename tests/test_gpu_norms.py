"""-m gpu: normalization(planes, 'gn' | 'in') of the drop-in modules (networks/unet.py; reference code/networks/unet.py:17-28) against
the fixture the reference's own Encoder(norm=..) / Decoder(norm=..) produced (tests/golden/modules_norm.npz): state_dict keys, forward
and every parameter gradient through torch autograd (train mode; nn.GroupNorm / nn.InstanceNorm2d are the same in eval
mode, checked below), the 16-image limit of the one-statistics-group-per-image scheme, and the parts that stay bn / dsbn only."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import modules_norm_states                  # noqa: E402

T = torch.from_numpy
DEV = 'cuda:0'


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def _models(M, norm):
    from networks.unet import Encoder, Decoder
    enc_sd, dec_sd = modules_norm_states(M, norm)
    enc, dec = Encoder(n=8, norm=norm).to(DEV), Decoder(n=8, num_classes=2, norm=norm).to(DEV)
    assert list(enc.state_dict().keys()) == list(M['%s.keys.enc' % norm])
    assert list(dec.state_dict().keys()) == list(M['%s.keys.dec' % norm])
    enc.load_state_dict(enc_sd); dec.load_state_dict(dec_sd)
    return enc, dec


@pytest.mark.parametrize('norm', ['gn', 'in'])
def test_gn_in_modules_forward_backward_vs_reference_fixture(golden_dir, norm):
    M = np.load(os.path.join(golden_dir, 'modules_norm.npz'))
    enc, dec = _models(M, norm)
    enc.train(); dec.train()
    x = T(M['x']).to(DEV)                                  # (the Encoder does not produce the image gradient: nothing in the path uses it)
    feats = enc(x)
    logits = dec(feats)
    for i, f in enumerate(feats):
        assert rel_l2(f.detach().cpu(), T(M['%s.feat%d' % (norm, i + 1)])) <= 2e-4, i
    assert rel_l2(logits.detach().cpu(), T(M['%s.logits' % norm])) <= 5e-4
    loss = (logits * T(M['%s.wl' % norm]).to(DEV)).sum() + (feats[2] * T(M['%s.wf' % norm]).to(DEV)).sum()
    loss.backward()
    worst = 0.0
    for nm, mod in (('enc', enc), ('dec', dec)):
        for k, p_ in mod.named_parameters():
            g = p_.grad.detach().cpu()
            ref_sig = M['%s.g%s.sig.%s' % (norm, nm, k)]
            if norm == 'in' and k.endswith('.bias') and '.conv' in k:
                assert float(g.abs().max()) == 0.0, k         # per-(image, channel) mean removed: exactly zero here
                continue
            fk = '%s.g%s.full.%s' % (norm, nm, k)
            if fk in M.files:
                r = rel_l2(g, T(M[fk]))
                worst = max(worst, r)
                assert r <= 2e-2, (k, r)
            np.testing.assert_allclose(float(g.double().norm()), np.sqrt(ref_sig[2]), rtol=2e-2, err_msg=k)
    assert worst > 0.0
    # eval mode: the statistics are still the input's (no running statistics exist)
    enc.eval(); dec.eval()
    with torch.no_grad():
        logits_e = dec(enc(T(M['x']).to(DEV)))
    assert rel_l2(logits_e.cpu(), T(M['%s.logits' % norm])) <= 5e-4


@pytest.mark.parametrize('norm', ['gn', 'in'])
def test_gn_in_batches_beyond_the_group_table_run_in_chunks(norm):
    """nn.GroupNorm / nn.InstanceNorm2d of the reference have no batch limit (code/networks/unet.py:20-23; --test_batch_size and the
    evaluation scripts' --batch_size are free); a launch plan holds 16 per-image statistics groups, so larger batches run as chunks of
    16 -- exact, because each image's statistics see that image only: forward and every gradient of a 21-image call equal the
    16- and 5-image calls."""
    from networks.unet import Encoder, Decoder
    torch.manual_seed(3)
    enc, dec = Encoder(n=8, norm=norm).to(DEV), Decoder(n=8, num_classes=2, norm=norm).to(DEV)
    enc.train(); dec.train()
    x = torch.randn(21, 3, 32, 32, device=DEV)
    wl = torch.randn(21, 2, 32, 32, device=DEV)

    def run(parts):
        for m in (enc, dec):
            m.zero_grad()
        outs = []
        for a, b in parts:
            lg = dec(enc(x[a:b]))
            (lg * wl[a:b]).sum().backward()
            outs.append(lg.detach())
        grads = [p.grad.detach().clone() for m in (enc, dec) for p in m.parameters() if p.grad is not None]
        return torch.cat(outs, 0), grads
    whole, g_whole = run([(0, 21)])
    split, g_split = run([(0, 16), (16, 21)])
    assert whole.shape == (21, 2, 32, 32) and torch.equal(whole, split)
    assert len(g_whole) == len(g_split) > 10
    for a, b in zip(g_whole, g_split):
        assert rel_l2(a.cpu(), b.cpu()) <= 1e-6 or float(b.abs().max()) == 0.0
    with torch.no_grad():
        assert torch.equal(dec(enc(x)), whole) or rel_l2(dec(enc(x)).cpu(), whole.cpu()) < 1e-6


def test_gn_in_limits_and_bn_only_parts():
    from networks.unet import ConvD, ConvU_Rec, Rec_Decoder, normalization
    import torch.nn as nn
    assert isinstance(normalization(8, 'gn'), nn.GroupNorm) and normalization(8, 'gn').num_groups == 1
    m = normalization(8, 'in')
    assert isinstance(m, nn.InstanceNorm2d) and not m.affine and not m.track_running_stats and len(m.state_dict()) == 0
    blk = ConvD(3, 8, 'gn', first=True).to(DEV)
    y = blk(torch.randn(16, 3, 16, 16, device=DEV))            # 16 images: the most ONE launch plan's group table holds
    assert y.shape == (16, 8, 16, 16) and torch.isfinite(y).all()
    for cls in (ConvU_Rec, ):
        with pytest.raises(NotImplementedError):
            cls(32, 'gn')
    with pytest.raises(NotImplementedError):
        Rec_Decoder(num_classes=3, norm='in')
