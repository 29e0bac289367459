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
            x = G[c + '.src'].astype(np.float32).copy()
            x /= 127.5
            x -= 1.0                                                     # fundus.py:217-218, float32
            np.testing.assert_array_equal(img[i], x.transpose(2, 0, 1), err_msg=c)   # bit-exact


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
        np.testing.assert_array_equal(img[i], oi)
    np.testing.assert_allclose(frq[2], img[2], atol=1e-6)        # lambda = 1 is the identity (then clip)
    # uint8 inputs (decoded PNG pixels, 1 byte per value): the same integers.  img bit for bit; the mixed image within the fp32 rounding of
    # the two row passes (uint8 rows run the matrix-core DFT of csrc/ram_dft.hip, fp32 rows the FFT)
    s8 = torch.from_numpy(np.stack(src).astype(np.uint8)).to(DEV)
    t8 = torch.from_numpy(np.stack(trg).astype(np.uint8)).to(DEV)
    img8, frq8 = R.source_to_target_freq_batch(s8, t8, torch.tensor(lam, dtype=torch.float32, device=DEV), 'fundus', torch.float32)
    np.testing.assert_array_equal(img8.cpu().numpy(), img)
    np.testing.assert_allclose(frq8.cpu().numpy(), frq, rtol=0, atol=2e-4 / 127.5)
    for i in range(B):
        np.testing.assert_allclose(frq8[i].cpu().numpy(), OR.ram_fundus(src[i], trg[i], lam[i])[1], rtol=0, atol=2e-3 / 127.5)


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


def test_matrix_core_row_pass_agrees_with_the_fft_row_pass():
    """uint8 images with coefficient tables run the row pass on the matrix cores (csrc/ram_dft.hip); without tables, and for fp32
    pixels, the FFT kernels: the same result within the fp32 rounding of either (2e-4 on the 0..255 scale), at every bench side, with
    rows that do not fill the last 32-row block and a batch of mixed ratios."""
    from ramdsir import _lib as L
    lib = L.lib()
    for S, H in ((400, 400), (256, 256), (384, 384), (512, 512), (400, 72), (320, 200)):
        rng = np.random.RandomState(S + H)
        B = 3
        src = torch.from_numpy(rng.randint(0, 256, (B, H, S, 3)).astype(np.uint8)).to(DEV)
        trg = torch.from_numpy(rng.randint(0, 256, (B, H, S, 3)).astype(np.uint8)).to(DEV)
        src[1, :, :, 2] = 0                                  # an all-zero channel: |F_src| == 0, the reference's angle() == 0 branch
        lam = torch.tensor([0.1, 0.5, 0.9], dtype=torch.float32, device=DEV)
        outs = []
        for tables in (True, False):
            m = R.RamMixer(B, H, S, torch.float32, DEV, 'fundus')
            assert m.tables is not None and int(lib.rd_ram_dft_tables_bytes(H, S, m.b)) == m.tables.numel()
            if not tables:
                m.p.dft_tables = None
            oi = torch.empty(B, H, S, 3, device=DEV)
            of = torch.empty(B, H, S, 3, device=DEV)
            m.bind(src, trg, lam, oi, of)
            m.run()
            torch.cuda.synchronize()
            outs.append((oi.cpu().numpy(), of.cpu().numpy()))
        np.testing.assert_array_equal(outs[0][0], outs[1][0])
        np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=0, atol=2e-4 / 127.5, err_msg=str((S, H)))
        assert np.abs(outs[0][1] - outs[0][0]).max() > 0.05          # the mix did something
    assert int(lib.rd_ram_dft_tables_bytes(400, 250, 25)) == 0       # a width the matrix pass does not take: the FFT kernels
    assert R.dft_tables(400, 250, 25, DEV) is None


def test_ram_rejects_unsupported_sizes():
    with pytest.raises(RuntimeError):
        _run([np.zeros((14, 14, 3), np.float32)], [np.zeros((14, 14, 3), np.float32)], [0.5])     # 14 = 2*7


def _lam_seeds():
    """The python-random seeds tests/golden/make_golden.py used so that the reference's own draw gives each ratio."""
    import random
    seeds = {}
    for sd in range(200):
        random.seed(sd)
        seeds.setdefault(random.randint(1, 10) / 10, sd)
    return seeds


def test_ram_trio_free_functions_match_reference_fixtures(golden_dir):
    """extract_amp_spectrum / low_freq_mutate_np / source_to_target_freq of the drop-in dataset modules (arrays in, arrays
    out, HIP underneath) against what the reference's numpy trio produced (fundus.py:13-61): ram.npz .amp_trg / .mutated /
    .freq_f64, incl. the zero-amplitude plane, lambda = 1, H != W and sides with factors 2, 3, 5."""
    import random
    from dataset import fundus as DF, prostate as DP
    assert DP.extract_amp_spectrum is DF.extract_amp_spectrum          # prostate.py:10-62 is the same trio
    G = np.load(os.path.join(golden_dir, 'ram.npz'))
    seeds = _lam_seeds()
    for c in G['cases']:
        src, trg, lam = G[c + '.src'], G[c + '.trg'], float(G[c + '.lam'])
        amp = DF.extract_amp_spectrum(trg.transpose(2, 0, 1))
        ref = G[c + '.amp_trg']
        assert isinstance(amp, np.ndarray) and amp.shape == ref.shape and amp.dtype == np.float32
        np.testing.assert_allclose(amp, ref, rtol=2e-5, atol=2e-6 * ref.max(), err_msg=c)
        a_s = np.abs(np.fft.fft2(src.astype(np.float64).transpose(2, 0, 1), axes=(-2, -1)))
        random.seed(seeds[lam])
        mut = DF.low_freq_mutate_np(a_s, ref, L=0.1)                   # draws the ratio itself, like fundus.py:35
        np.testing.assert_allclose(mut, G[c + '.mutated'], rtol=1e-5, atol=1e-6 * ref.max(), err_msg=c)
        random.seed(seeds[lam])
        frq = DF.source_to_target_freq(src, ref, L=0.1)
        assert frq.shape == src.shape
        np.testing.assert_allclose(frq, G[c + '.freq_f64'], rtol=0, atol=2e-3, err_msg=c)


def test_dataset_pieces_through_gpu_ram_match_reference_img_freq(golden_dir, tmp_path):
    """End of the R4 chain on the GPU: the drop-in Fundus_Multi's pieces (uint8 image, uint8 partner, lambda) mixed by
    rd_ram_mix == the img_freq the REFERENCE's Fundus_Multi.__getitem__ returned under the same seeds (sampling.npz)."""
    import random
    import synth_data as SD
    from dataset.fundus import Fundus_Multi, ram_collate
    import dataset.transform as trans
    G = np.load(os.path.join(golden_dir, 'sampling.npz'))
    base = SD.make_fundus_tree(str(tmp_path))
    tf = [trans.Resize((256, 256)), trans.RandomScaleCrop((256, 256))]

    def compose(sm):
        for t in tf:
            sm = t(sm)
        return sm
    ds = Fundus_Multi(domain_idx_list=[1], base_dir=base, split='train', transform=compose, is_out_domain=True, test_domain_idx=0)
    random.seed(1337)
    np.random.seed(1337)
    pieces = [ds[i] for i in range(int(G['f_ood.n']))]
    src = torch.stack([p[0] for p in pieces]).to(DEV)
    trg = torch.stack([p[1] for p in pieces]).to(DEV)
    lam = torch.stack([p[2] for p in pieces]).to(DEV)
    assert src.dtype == torch.uint8
    img, frq = ram_collate(src, trg, lam, 'fundus')
    for i in range(len(pieces)):
        np.testing.assert_array_equal(img[i].cpu().numpy()[:, ::8, ::8], G['f_ood.%d.img' % i])
        assert np.abs(frq[i].cpu().numpy()[:, ::8, ::8] - G['f_ood.%d.frq' % i]).max() < 2e-3 / 127.5, i
