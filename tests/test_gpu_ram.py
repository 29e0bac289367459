"""-m gpu: the RAM FFT kernels against the fixtures produced by the reference's own numpy trio and the
oracle at the benchmark size.  Tolerance: 2e-3 on the 0..255 scale for the fp32 path (SURVEY.md section 7:
the reference's own float32-vs-float64 numpy spread is up to 9.5e-4), i.e. 1.6e-5 after /127.5."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ramdsir import ram as R       # noqa: E402
from oracle import ram as OR      # noqa: E402

DEV = 'cuda:0'


def _run(src, trg, lam, dataset='fundus', dtype=torch.float32):
    s = torch.from_numpy(np.stack(src)).to(DEV)
    t = torch.from_numpy(np.stack(trg)).to(DEV)
    l = torch.tensor(lam, dtype=torch.float32, device=DEV)
    img, frq = R.source_to_target_freq_batch(s, t, l, dataset, dtype)
    torch.cuda.synchronize()
    return img.cpu().numpy(), frq.cpu().numpy()


def test_ram_matches_reference_fixtures(golden_dir):
    G = np.load(os.path.join(golden_dir, 'ram.npz'))
    by_shape = {}
    for c in G['cases']:
        by_shape.setdefault(G[c + '.src'].shape, []).append(c)
    assert len(by_shape) >= 8
    for shape, cases in by_shape.items():                     # one batched launch per geometry, mixed lambdas
        img, frq = _run([G[c + '.src'] for c in cases], [G[c + '.trg'] for c in cases], [float(G[c + '.lam']) for c in cases])
        for i, c in enumerate(cases):
            ref = np.clip(G[c + '.freq_f64'], 0, 255) / 127.5 - 1.0
            np.testing.assert_allclose(frq[i], ref.transpose(2, 0, 1), rtol=0, atol=2e-3 / 127.5, err_msg=c)
            np.testing.assert_allclose(img[i], (G[c + '.src'] / 127.5 - 1.0).transpose(2, 0, 1), rtol=0, atol=1e-6)


@pytest.mark.parametrize('S', [256, 384, 400, 512])
def test_ram_benchmark_sizes_vs_oracle(S):
    rng = np.random.RandomState(S)
    B = 3
    src = [np.round(rng.uniform(0, 255, (S, S, 3))).astype(np.float32) for _ in range(B)]
    trg = [np.round(rng.uniform(0, 255, (S, S, 3))).astype(np.float32) for _ in range(B)]
    lam = [0.1, 0.6, 1.0]
    img, frq = _run(src, trg, lam)
    for i in range(B):
        oi, of = OR.ram_fundus(src[i], trg[i], lam[i])
        np.testing.assert_allclose(frq[i], of, rtol=0, atol=2e-3 / 127.5)
        np.testing.assert_allclose(img[i], oi, rtol=0, atol=1e-6)
    np.testing.assert_allclose(frq[2], img[2], atol=1e-6)        # lambda = 1 is the identity (then clip)


def test_ram_prostate_call_site_and_bf16_output():
    rng = np.random.RandomState(7)
    S, B = 384, 2
    src = [rng.uniform(-1, 1, (S, S, 3)).astype(np.float32) for _ in range(B)]
    trg = [rng.uniform(-1, 1, (S, S, 3)).astype(np.float32) for _ in range(B)]
    lam = [0.2, 0.9]
    img, frq = _run(src, trg, lam, 'prostate')
    for i in range(B):
        oi, of = OR.ram_prostate(src[i], trg[i], lam[i])
        np.testing.assert_allclose(frq[i], of, rtol=0, atol=2e-5)
        assert frq[i].min() >= -1 and frq[i].max() <= 1
    imgb, frqb = _run(src, trg, lam, 'prostate', torch.bfloat16)
    np.testing.assert_allclose(frqb, frq, rtol=0, atol=2 ** -8)    # bf16 storage of values in [-1,1]


def test_ram_rejects_unsupported_sizes():
    with pytest.raises(RuntimeError):
        _run([np.zeros((14, 14, 3), np.float32)], [np.zeros((14, 14, 3), np.float32)], [0.5])     # 14 = 2*7
