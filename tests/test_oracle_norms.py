"""CPU: the oracle's gn / in restatement (oracle/unet.py, normalization(planes, 'gn' | 'in') of code/networks/unet.py:20-23) against
the fixture the reference's own Encoder(norm=..) / Decoder(norm=..) produced (tests/golden/modules_norm.npz, make_golden.py
gen_modules_norm): state_dict keys, forward, input / parameter gradients."""
import os

import numpy as np
import pytest
import torch

from golden_util import assert_sig_close, modules_norm_states, state_checksum
from oracle import unet as OU

T = torch.from_numpy


@pytest.mark.parametrize('norm', ['gn', 'in'])
def test_oracle_gn_in_modules_vs_reference_fixture(golden_dir, norm):
    M = np.load(os.path.join(golden_dir, 'modules_norm.npz'))
    enc, dec = modules_norm_states(M, norm)
    np.testing.assert_allclose(state_checksum(enc), M['%s.chk.enc' % norm], rtol=1e-12)
    np.testing.assert_allclose(state_checksum(dec), M['%s.chk.dec' % norm], rtol=1e-12)
    assert list(enc.keys()) == list(M['%s.keys.enc' % norm]) and list(dec.keys()) == list(M['%s.keys.dec' % norm])
    enc, dec = OU.clone_state(enc, requires_grad=True), OU.clone_state(dec, requires_grad=True)
    x = T(M['x']).clone().requires_grad_(True)
    feats = OU.encoder_forward(x, enc, True)
    logits = OU.decoder_forward(feats, dec, True)
    for i, f in enumerate(feats):
        ref = M['%s.feat%d' % (norm, i + 1)]
        np.testing.assert_allclose(f.detach().numpy(), ref, rtol=1e-3, atol=1e-4 * np.abs(ref).max())
    ref = M['%s.logits' % norm]
    np.testing.assert_allclose(logits.detach().numpy(), ref, rtol=1e-3, atol=2e-4 * np.abs(ref).max())
    ((logits * T(M['%s.wl' % norm])).sum() + (feats[2] * T(M['%s.wf' % norm])).sum()).backward()
    ref = M['%s.dx' % norm]
    np.testing.assert_allclose(x.grad.numpy(), ref, rtol=1e-2, atol=1e-3 * np.abs(ref).max())
    for nm, sd in (('enc', enc), ('dec', dec)):
        for k in OU.param_keys(sd):
            g = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
            if norm == 'in' and k.endswith('.bias') and '.conv' in k:
                continue                # InstanceNorm removes the per-(image, channel) mean: analytically zero, fp32 noise in the fixture
            assert_sig_close(g, M['%s.g%s.sig.%s' % (norm, nm, k)], 2e-3, name=nm + '.' + k)
