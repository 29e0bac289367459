"""CPU: the drop-in datasets / transforms / Dice metric against fixtures produced by the REFERENCE's own code
(tests/golden/make_golden.py gen_sampling / gen_metrics: Fundus_Multi fundus.py:160-240, Prostate_Multi
prostate.py:152-202, transform.py:16-44,163-204, utils/metrics.py:55-109) on the synthetic trees of
tests/synth_data.py under seeds 1337: every random draw in order, the source image, the mask, and -- through the
oracle's numpy RAM on the pieces the drop-in returns -- the reference's img_freq."""
import os
import random

import numpy as np
import pytest
import torch

import synth_data as SD
from golden_util import sig
from oracle import ram as OR

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


class Compose:
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, s):
        for t in self.ts:
            s = t(s)
        return s


@pytest.fixture(scope='module')
def trees(tmp_path_factory):
    root = str(tmp_path_factory.mktemp('trees'))
    return SD.make_fundus_tree(root), SD.make_prostate_tree(root)


@pytest.mark.parametrize('tag,dom,ood,tdi', [('f_ood', [1], True, 0), ('f_ind', [2, 3], False, 1)])
def test_fundus_multi_sampling_matches_reference(trees, tag, dom, ood, tdi):
    from dataset.fundus import Fundus_Multi
    import dataset.transform as trans
    G = np.load(os.path.join(GOLD, 'sampling.npz'))
    tf = Compose([trans.Resize((256, 256)), trans.RandomScaleCrop((256, 256))])          # train.py:541
    ds = Fundus_Multi(domain_idx_list=dom, base_dir=trees[0], split='train', transform=tf, is_out_domain=ood, test_domain_idx=tdi)
    random.seed(1337)
    np.random.seed(1337)
    for i in range(int(G[tag + '.n'])):
        with SD.DrawLog() as dl:
            img, other, lam, mask = ds[i]
        assert dl.log == list(G['%s.%d.draws' % (tag, i)]), i                            # same draws, same order
        assert img.shape == (256, 256, 3) and other.shape == (256, 256, 3) and img.dtype == torch.uint8 and other.dtype == torch.uint8
        x = img.numpy().astype(np.float32)                                                # fundus.py:211,217-218
        x /= 127.5
        x -= 1.0
        x = x.transpose(2, 0, 1)
        np.testing.assert_array_equal(x[:, ::8, ::8], G['%s.%d.img' % (tag, i)])         # bit-exact source image
        np.testing.assert_array_equal(mask.numpy()[:, ::4, ::4].astype(np.uint8), G['%s.%d.mask' % (tag, i)])
        s = G['%s.%d.sig' % (tag, i)]
        np.testing.assert_allclose(sig(torch.from_numpy(x))[:4], s[0][:4], rtol=1e-12)
        np.testing.assert_allclose(sig(mask)[:4], s[2][:4], rtol=0)
        # RAM on the returned pieces (oracle, float32 like numpy >= 2 computes it) == the reference's img_freq
        _, frq = OR.ram_fundus(img.numpy().astype(np.float32), other.numpy().astype(np.float32), float(lam), dtype=np.float32)
        assert np.abs(frq[:, ::8, ::8] - G['%s.%d.frq' % (tag, i)]).max() < 2e-3 / 127.5
        np.testing.assert_allclose(sig(torch.from_numpy(frq))[2], s[1][2], rtol=1e-5)


def test_prostate_multi_sampling_matches_reference(trees, monkeypatch):
    from dataset.prostate import Prostate_Multi
    G = np.load(os.path.join(GOLD, 'sampling.npz'))
    real = os.listdir
    monkeypatch.setattr(os, 'listdir', lambda p: sorted(real(p)))          # the fixture was drawn with sorted listings
    ds = Prostate_Multi(domain_idx_list=[0, 2], base_dir=trees[1], split='train', is_out_domain=True, test_domain_idx=4)
    random.seed(1337)
    np.random.seed(1337)
    for i in range(int(G['p_ood.n'])):
        with SD.DrawLog() as dl:
            img, other, lam, mask = ds[i]
        assert dl.log == list(G['p_ood.%d.draws' % i]), i
        x = img.numpy().transpose(2, 0, 1)
        np.testing.assert_array_equal(x[:, ::4, ::4], G['p_ood.%d.img' % i])
        _, frq = OR.ram_prostate(img.numpy(), other.numpy(), float(lam), dtype=np.float32)
        assert np.abs(frq[:, ::4, ::4] - G['p_ood.%d.frq' % i]).max() < 2e-5
        s = G['p_ood.%d.sig' % i]
        np.testing.assert_allclose(sig(mask.float())[:4], s[2][:4], rtol=0)


def test_random_crop_always_draws_and_pads_like_the_reference():
    """transform.py:16-44: right/bottom padding (image 0, mask 255) for small inputs, and two randint draws even when
    the image already has the crop size."""
    from PIL import Image
    import dataset.transform as trans
    img = Image.fromarray(np.full((20, 30, 3), 7, np.uint8))
    msk = Image.fromarray(np.full((20, 30), 9, np.uint8))
    random.seed(5)
    with SD.DrawLog() as dl:
        out = trans.RandomCrop((32, 32))({'img': img, 'mask': msk})
    assert dl.log == ['randint(0,0):0', 'randint(0,0):0']
    a, m = np.array(out['img']), np.array(out['mask'])
    assert a.shape == (32, 32, 3) and (a[:20, :30] == 7).all() and (a[20:] == 0).all() and (a[:, 30:] == 0).all()
    assert (m[:20, :30] == 9).all() and (m[20:] == 255).all() and (m[:, 30:] == 255).all()
    r = trans.Resize((24, 16))({'img': img, 'mask': msk})                                # (w, h)
    assert r['img'].size == (24, 16) and r['mask'].size == (24, 16)


def test_dice_metrics_match_reference_fixture():
    from utils import metrics as M
    G = np.load(os.path.join(GOLD, 'metrics.npz'))
    for i in range(int(G['ncases'])):
        got = M.dice_coefficient_numpy(G['d%d.pred' % i], G['d%d.gt' % i])
        assert got == float(G['d%d.dice' % i]), i                                        # same float64 arithmetic: exact
    pred, gt = G['b.pred'].astype(np.float32), torch.from_numpy(G['b.gt'].astype(np.float32))
    assert tuple(M.dice_coeff_2label(pred[0], gt[0])) == tuple(G['b.single'])
    np.testing.assert_allclose(M.dice_coeff_2label(pred, gt), G['b.batch'], rtol=1e-15)
    a, b = torch.from_numpy(G['t.a']), torch.from_numpy(G['t.b'])
    np.testing.assert_allclose(M.dice(a, b).item(), float(G['t.dice']), rtol=1e-7)
    li, lt = torch.from_numpy(G['m.a']), torch.from_numpy(G['m.b'])
    np.testing.assert_allclose(float(M.dice_multi(li, lt, 3)), float(G['m.dice_multi']), rtol=1e-7)
    np.testing.assert_allclose(float(M.dice_multi(li, lt, 3, ignore_index=0)), float(G['m.dice_multi_ign0']), rtol=1e-7)
