"""-m gpu: the drop-in nn.Modules (networks.unet.Encoder / Decoder / Rec_Decoder) against the fixtures the
reference's own modules produced (tests/golden/modules.npz): forward in train and eval mode, running
statistics, parameter gradients through torch autograd, state_dict round trip."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import modules_states, bn_shadowed_bias      # noqa: E402

T = torch.from_numpy
DEV = 'cuda:0'


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def _models():
    from networks.unet import Encoder, Decoder, Rec_Decoder
    enc_sd, dec_sd, rec_sd, _ = modules_states()
    enc, dec = Encoder().to(DEV), Decoder(num_classes=2).to(DEV)
    rec = Rec_Decoder(num_classes=3, norm='dsbn', num_domains=3).to(DEV)
    enc.load_state_dict(enc_sd); dec.load_state_dict(dec_sd); rec.load_state_dict(rec_sd)
    return enc, dec, rec


def test_state_dict_keys_and_module_surface(golden_dir):
    import torch.nn as nn
    from networks.unet import Encoder, Decoder, Rec_Decoder, count_params, normalization
    with open(os.path.join(golden_dir, 'state_manifest.json')) as f:
        man = json.load(f)
    for nm, m in (('encoder', Encoder()), ('seg_decoder', Decoder(num_classes=2)),
                  ('rec_decoder', Rec_Decoder(num_classes=3, norm='dsbn', num_domains=3))):
        got = [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]
        assert got == man[nm]
        assert sum(p.numel() for p in m.parameters()) == man[nm + '_params']
    assert abs(count_params(Encoder()) - 1.967904) < 1e-9
    assert sum(isinstance(m, nn.BatchNorm2d) for m in Encoder().modules()) == 15
    assert sum(isinstance(m, nn.Conv2d) for m in Decoder().modules()) == 12
    with pytest.raises(ValueError):
        normalization(8, 'nope')
    from networks.dsbn import DomainSpecificBatchNorm2d
    with pytest.raises(ValueError):
        DomainSpecificBatchNorm2d(4, 2)(torch.zeros(2, 4, 3), torch.zeros(2, dtype=torch.long))


@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_modules_forward_backward_vs_reference_fixture(golden_dir, mode):
    M = np.load(os.path.join(golden_dir, 'modules.npz'))
    enc, dec, rec = _models()
    for m in (enc, dec, rec):
        m.train() if mode == 'train' else m.eval()
    x = T(M['x']).to(DEV)
    feats = enc(x)
    logits = dec(feats)
    r1 = rec(feats[-1][0:2], domain_label=1 * torch.ones(2, dtype=torch.long))
    r2 = rec(feats[-1][2:4], domain_label=2 * torch.ones(2, dtype=torch.long))
    for i, f in enumerate(feats):
        assert rel_l2(f.cpu(), T(M['%s.feat%d' % (mode, i + 1)])) < 2e-5, i
    for got, key in ((logits, 'logits'), (r1, 'rec_d1'), (r2, 'rec_d2')):
        assert rel_l2(got.detach().cpu(), T(M['%s.%s' % (mode, key)])) < 2e-4, key
    if mode == 'eval':
        return
    loss = (logits * T(M['wl']).to(DEV)).sum() + (r1 * T(M['wr0']).to(DEV)).sum() + (r2 * T(M['wr1']).to(DEV)).sum()
    loss.backward()
    # Gradient agreement with the reference fixture is limited by DISCRETE events, not by arithmetic precision: the order of
    # the fp32 atomics that accumulate the BatchNorm sums varies run to run, the last bit of a batch statistic with it, and
    # on this 32x32 fixture (2x2-pixel, 2-image batches at the bottleneck) that flips individual ReLU kinks / max-pool ties.
    # scripts/mod_noise.py, 40 runs on MI355X: 3 runs match to 6e-5 per tensor (2e-5 aggregate), most sit at 1.9e-2 per
    # tensor (the same flip every time), the tail reached 6.1e-2 per tensor / 2.4e-2 aggregate.  Hence: every tensor within
    # 1e-1, and the relative L2 error over ALL parameter gradients together within 4e-2.
    num = den = 0.0
    for nm, m in (('enc', enc), ('dec', dec), ('rec', rec)):
        for k, p in m.named_parameters():
            g = p.grad.cpu() if p.grad is not None else torch.zeros(p.shape)
            if bn_shadowed_bias(k):
                assert float(g.abs().max()) == 0.0
                continue
            ref = M['train.g%s.sig.%s' % (nm, k)]
            np.testing.assert_allclose(float(g.double().norm()), np.sqrt(ref[2]), rtol=1e-1, atol=1e-9, err_msg=k)
            fk = 'train.g%s.full.%s' % (nm, k)
            if fk in M.files and np.abs(M[fk]).max() > 0:
                r = T(M[fk]).double()
                assert rel_l2(g, r) < 1e-1, k
                num += float((g.double() - r).pow(2).sum())
                den += float(r.pow(2).sum())
    assert den > 0 and (num / den) ** 0.5 < 4e-2, (num / den) ** 0.5
    for nm, m in (('enc', enc), ('dec', dec), ('rec', rec)):
        for k, v in m.state_dict().items():
            if 'running' in k or 'num_batches' in k:
                np.testing.assert_allclose(v.cpu().numpy(), M['train.buf.%s.%s' % (nm, k)], rtol=1e-4, atol=1e-6, err_msg=k)


def test_bn_train_mode_inference_like_test_fundus_slice():
    """test_fundus_slice.py:75-83: model.eval() then every BatchNorm2d back to .train() -> batch statistics."""
    import torch.nn as nn
    enc, dec, _ = _models()
    enc.eval(); dec.eval()
    x = torch.randn(4, 3, 32, 32, device=DEV)
    with torch.no_grad():
        y_eval = dec(enc(x))
        for m in list(enc.modules()) + list(dec.modules()):
            if isinstance(m, nn.BatchNorm2d):
                m.train()
        y_bn = dec(enc(x))
    assert y_eval.shape == y_bn.shape == (4, 2, 32, 32)
    assert float((y_eval - y_bn).abs().max()) > 1e-3          # the two modes genuinely differ
    assert int(enc.convd1.bn1.num_batches_tracked) == 1       # and train-mode BN tracked the batch
