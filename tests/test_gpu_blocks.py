"""-m gpu: the module surface SURVEY.md 8b lists BELOW Encoder / Decoder / Rec_Decoder, each called on its own on the
HIP engine (fp32 storage) against fixtures produced by the reference's own modules:
  ConvD / ConvU / ConvU_Rec .forward + backward   code/networks/unet.py:52-72, 96-117, 139-165   tests/golden/blocks.npz
  DomainSpecificBatchNorm2d.forward -> (y, label) code/networks/dsbn.py:24-27                    tests/golden/dsbn.npz
Tolerances: outputs 1e-4 relative; gradients 2e-3 of each tensor's max (BN batches are 2-3 images of 8x8..16x16 pixels);
running statistics 1e-5; conv biases in front of a train-mode BatchNorm: exact zero here (fp32 noise in the reference)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from networks import unet as NU                                  # noqa: E402
from networks.dsbn import DomainSpecificBatchNorm2d              # noqa: E402

DEV = 'cuda:0'
T = torch.from_numpy


@pytest.fixture(scope='module')
def B(golden_dir):
    return np.load(os.path.join(golden_dir, 'blocks.npz'))


def _load(mod, G, tag):
    sd = {k[len(tag) + 4:]: T(G[k]) for k in G.files if k.startswith(tag + '.sd.')}
    mod.load_state_dict(sd, strict=True)
    return mod.to(DEV).train()


def _run_block(G, tag, mod, kwargs=None):
    mod = _load(mod, G, tag)
    ins = [T(G['%s.in%d' % (tag, i)]).to(DEV).requires_grad_(True) for i in range(2) if '%s.in%d' % (tag, i) in G.files]
    y = mod(*ins, **(kwargs or {}))
    ref = G[tag + '.y']
    assert y.shape == ref.shape and y.dtype == torch.float32
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
    (y * T(G[tag + '.w']).to(DEV)).sum().backward()
    for i, t in enumerate(ins):
        ref = G['%s.din%d' % (tag, i)]
        np.testing.assert_allclose(t.grad.cpu().numpy(), ref, rtol=1e-3, atol=2e-3 * np.abs(ref).max(), err_msg='din%d' % i)
    for k, p in mod.named_parameters():
        ref = G['%s.g.%s' % (tag, k)]
        g = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
        if 'conv' in k and k.endswith('.bias'):
            assert np.abs(g).max() == 0.0, k
            continue
        np.testing.assert_allclose(g, ref, rtol=1e-3, atol=2e-3 * (np.abs(ref).max() + 1e-3), err_msg=k)
    for k, v in mod.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            np.testing.assert_allclose(v.cpu().numpy(), G['%s.after.%s' % (tag, k)], rtol=1e-5, atol=1e-6, err_msg=k)
    return mod


def test_convd_forward_backward_on_its_own(B):
    _run_block(B, 'convd_first', NU.ConvD(3, 16, 'bn', first=True))
    _run_block(B, 'convd', NU.ConvD(16, 32, 'bn'))                       # 2x2 max-pool of a raw input fused into conv1's read
    _run_block(B, 'convd_leaky', NU.ConvD(16, 32, 'bn', activation='leaky'))


def test_convu_forward_backward_on_its_own(B):
    _run_block(B, 'convu_first', NU.ConvU(64, 'bn', first=True))
    _run_block(B, 'convu', NU.ConvU(32, 'bn'))


def test_convu_rec_forward_backward_on_its_own_picks_the_domain(B):
    m = _run_block(B, 'convu_rec', NU.ConvU_Rec(64, 'dsbn', num_domains=3), dict(domain_label=2 * torch.ones(3, dtype=torch.long)))
    assert int(m.bn1.bns[0].num_batches_tracked) == 0 and int(m.bn1.bns[2].num_batches_tracked) == 1


def test_block_inside_encoder_still_runs_after_a_standalone_call():
    """A block called on its own re-homes its parameters into its own arena; the parent's next call must pick them up again."""
    torch.manual_seed(0)
    enc = NU.Encoder().to(DEV).train()
    x = torch.randn(2, 3, 64, 64, device=DEV)                       # 4x4-pixel bottleneck: BN batches of 32 values, not 8
    with torch.no_grad():
        f0 = [f.clone() for f in enc(x)]
        y = enc.convd1(x)
        np.testing.assert_allclose(y.cpu().numpy(), f0[0].cpu().numpy(), rtol=1e-4, atol=1e-5)
        f1 = enc(x)
    for a, b in zip(f0, f1):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-3, atol=1e-4)   # atomics order, amplified by the small deep-level BN batches
    assert int(enc.convd1.bn1.num_batches_tracked) == 3 and int(enc.convd5.bn1.num_batches_tracked) == 2


def test_dsbn_forward_returns_tuple_and_matches_reference(golden_dir):
    G = np.load(os.path.join(golden_dir, 'dsbn.npz'))
    m = DomainSpecificBatchNorm2d(12, num_domains=3)
    m.load_state_dict({k[3:]: T(G[k]) for k in G.files if k.startswith('sd.')}, strict=True)
    m = m.to(DEV).train()
    x = T(G['x']).to(DEV).requires_grad_(True)
    lab = T(G['lab'])                                                      # CPU int64 tensor, as train.py:268 passes it
    y, lab_out = m(x, lab)
    assert lab_out is lab
    np.testing.assert_allclose(y.detach().cpu().numpy(), G['y'], rtol=1e-4, atol=2e-5)      # incl. the |mean| = 1e3 sigma channel
    (y * T(G['w']).to(DEV)).sum().backward()
    ref = G['dx']
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref, rtol=1e-3, atol=1e-3 * np.abs(ref).max())
    for k, p in m.named_parameters():
        ref = G['g.' + k]
        g = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
        np.testing.assert_allclose(g, ref, rtol=1e-3, atol=1e-3 * (np.abs(ref).max() + 1e-3), err_msg=k)
    for k, v in m.state_dict().items():
        if 'running' in k or 'num_batches' in k:
            np.testing.assert_allclose(v.cpu().numpy(), G['after.' + k], rtol=1e-5, atol=1e-6, err_msg=k)
    m.eval()
    y2, _ = m(T(G['x']).to(DEV), lab)
    np.testing.assert_allclose(y2.detach().cpu().numpy(), G['y_eval'], rtol=1e-4, atol=2e-5)
    with pytest.raises(ValueError):
        m(torch.zeros(3, 12, device=DEV), lab)                             # dsbn.py:30-34
