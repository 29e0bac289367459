"""Helpers shared by the oracle tests and the GPU parity tests."""
import json
import os

import numpy as np
import torch

from oracle import unet as OU


def sig(t):
    a = t.detach().double().reshape(-1)
    return np.array([a.sum().item(), a.abs().sum().item(), (a * a).sum().item(), a.numel()]
                    + a[:8].tolist() + [0.0] * max(0, 8 - a.numel()), dtype=np.float64)


def state_checksum(sd):
    return np.array([sum(v.double().sum().item() for v in sd.values()),
                     sum(v.double().abs().sum().item() for v in sd.values())])


def assert_sig_close(t, ref_sig, rtol, name=''):
    """Compare a tensor with a stored signature: L2 norm, abs-sum and the first 8 elements."""
    s = sig(t)
    assert s[3] == ref_sig[3], name
    scale = np.sqrt(ref_sig[2] / max(ref_sig[3], 1)) + 1e-30          # RMS of the reference tensor
    n = int(min(8, ref_sig[3]))
    np.testing.assert_allclose(s[4:4 + n], ref_sig[4:4 + n], rtol=0, atol=rtol * scale * 4 + 1e-12, err_msg=name)
    np.testing.assert_allclose(np.sqrt(s[2]), np.sqrt(ref_sig[2]), rtol=rtol, atol=1e-12, err_msg=name)
    np.testing.assert_allclose(s[1], ref_sig[1], rtol=rtol, atol=1e-12, err_msg=name)


def modules_states():
    """The exact states tests/golden/make_golden.py:gen_modules loaded into the reference modules."""
    enc_sd, dec_sd, rec_sd = OU.encoder_state(seed=10), OU.decoder_state(seed=11), OU.rec_decoder_state(seed=12)
    chks = tuple(map(state_checksum, (enc_sd, dec_sd, rec_sd)))
    g = torch.Generator().manual_seed(99)
    for sd in (enc_sd, dec_sd, rec_sd):
        for k in sd:
            if ('.bn' in k) and k.endswith('.weight'):
                sd[k] = 1.0 + 0.2 * torch.randn(sd[k].shape, generator=g)
            if ('.bn' in k) and k.endswith('.bias'):
                sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    return enc_sd, dec_sd, rec_sd, chks


def step_states(meta):
    nd = len(meta['batch_sizes'])
    enc_sd = OU.encoder_state(seed=20)
    dec_sd = OU.decoder_state(num_classes=meta['num_classes'], seed=21)
    rec_sd = OU.rec_decoder_state(num_classes=3, num_domains=nd, seed=22)
    return enc_sd, dec_sd, rec_sd


def load_step(golden_dir, name):
    G = np.load(os.path.join(golden_dir, 'step_%s.npz' % name))
    meta = json.loads(str(G['meta']))
    return G, meta


def bn_shadowed_bias(key):
    """conv biases that feed a train-mode BN: their gradient is analytically 0 (the reference holds
    fp32 cancellation noise there), so they are checked against a noise bound, not element-wise."""
    return key.endswith('.bias') and '.conv' in key


def assert_noise_level(g, weight_sig, name=''):
    wrms = np.sqrt(weight_sig[2] / max(weight_sig[3], 1))
    assert float(g.detach().abs().max()) <= 1e-3 * (wrms + 1.0) + 1e-6, name


def modules_norm_states(M, norm):
    """The states tests/golden/make_golden.py:gen_modules_norm loaded into the reference's Encoder(n=8, norm) / Decoder(n=8, norm):
    generated conv weights (same seeds), the affine parameters of the fixture."""
    enc_sd, dec_sd = OU.encoder_state(n=8, seed=20, norm=norm), OU.decoder_state(n=8, seed=21, norm=norm)
    for nm, sd in (('enc', enc_sd), ('dec', dec_sd)):
        for k in sd:
            if '.bn' in k:
                sd[k] = torch.from_numpy(M['%s.sd.%s.%s' % (norm, nm, k)]).clone()
    return enc_sd, dec_sd
