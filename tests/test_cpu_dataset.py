"""CPU: host input pipeline semantics of the drop-in datasets (no GPU work happens in __getitem__)."""
import os
import random

import numpy as np
import torch
from PIL import Image

from oracle import masks as OM


def _tree(root, size=48, n=5):
    rng = np.random.RandomState(1)
    base = os.path.join(root, 'fundus')
    for d in range(1, 5):
        os.makedirs(os.path.join(base, 'Domain%d' % d, 'train', 'ROIs', 'image'), exist_ok=True)
        os.makedirs(os.path.join(base, 'Domain%d' % d, 'train', 'ROIs', 'mask'), exist_ok=True)
        lines, partner = [], []
        for i in range(n):
            img = np.full((size, size, 3), 10 * d + i, np.uint8)               # pixel value encodes the domain
            m = rng.choice([0, 50, 51, 128, 200, 201, 255], size=(size, size)).astype(np.uint8)
            ri, rm = 'Domain%d/train/ROIs/image/%d.png' % (d, i), 'Domain%d/train/ROIs/mask/%d.png' % (d, i)
            Image.fromarray(img).save(os.path.join(base, ri))
            Image.fromarray(m).save(os.path.join(base, rm))
            lines.append(ri + ' ' + rm)
            partner.append('train/ROIs/image/%d.png train/ROIs/mask/%d.png' % (i, i))
        open(os.path.join(base, 'Domain%d_train.list' % d), 'w').write('\n'.join(lines) + '\n')
        open(os.path.join(base, 'Domain%d' % d, 'train.list'), 'w').write('\n'.join(partner) + '\n')
    return base


def test_fundus_multi_returns_ram_pieces_with_reference_sampling(tmp_path):
    from dataset.fundus import Fundus_Multi
    base = _tree(str(tmp_path))
    ds = Fundus_Multi(domain_idx_list=[1], base_dir=base, split='train', transform=None, is_out_domain=True, test_domain_idx=0)
    assert len(ds) == 5
    random.seed(3); np.random.seed(3)
    seen = set()
    for k in range(40):
        img, other, lam, mask = ds[k % 5]
        assert img.shape == (48, 48, 3) and img.dtype == torch.uint8 and other.shape == (48, 48, 3) and other.dtype == torch.uint8
        assert mask.shape == (2, 48, 48)
        assert round(float(lam) * 10) in range(1, 11)
        dom = int(round(float(other[0, 0, 0]))) // 10
        seen.add(dom)
        assert 20 <= float(img[0, 0, 0]) < 30                       # own domain = Domain2
    assert seen == {3, 4}                                             # never the test domain (1), never its own (2)
    ds2 = Fundus_Multi(domain_idx_list=[1], base_dir=base, split='train', transform=None, is_out_domain=False, test_domain_idx=0)
    doms = {int(round(float(ds2[0][1][0, 0, 0]))) // 10 for _ in range(40)}
    assert doms == {2, 3, 4}


def test_mask_encoding_matches_oracle(tmp_path):
    from dataset.transform import fundus_mask
    g = np.array([[0, 50, 51, 128], [200, 201, 255, 49]], np.uint8)
    np.testing.assert_array_equal(fundus_mask(g), OM.fundus_mask_multilabel(g))


def test_transforms_shapes_and_ranges():
    from dataset.transform import Resize, RandomScaleCrop, Normalize
    rng = np.random.RandomState(0)
    s = {'img': Image.fromarray(rng.randint(0, 255, (80, 100, 3)).astype(np.uint8)),
         'mask': Image.fromarray(rng.choice([0, 128, 255], size=(80, 100)).astype(np.uint8))}
    random.seed(0)
    s = RandomScaleCrop((64, 64))(Resize((64, 64))(s))
    assert s['img'].size == (64, 64) and s['mask'].size == (64, 64)
    n = Normalize()(s)
    assert n['img'].shape == (3, 64, 64) and float(n['img'].min()) >= -1 and float(n['img'].max()) <= 1
    assert n['mask'].shape == (2, 64, 64)


def test_metrics_postprocessing():
    from utils.metrics import dice_coefficient_numpy, get_largest_fillhole, postprocessing
    a = np.zeros((10, 10), np.uint8); a[1:6, 1:6] = 1; a[3, 3] = 0; a[8, 8] = 1          # big blob with a hole + a speck
    out = get_largest_fillhole(a.copy())
    assert out[3, 3] and not out[8, 8] and out.sum() == 25
    assert abs(dice_coefficient_numpy(out, out) - 1.0) < 1e-12
    assert dice_coefficient_numpy(np.zeros((4, 4)), np.zeros((4, 4))) == 1.0              # +1 smoothing
    p = np.stack([a * 0.9, a * 0.9]).astype(np.float32)
    assert postprocessing(torch.from_numpy(p), threshold=0.75).shape == (2, 10, 10)


def test_prostate_multi_pieces(tmp_path):
    from dataset.prostate import Prostate_Multi
    rng = np.random.RandomState(2)
    base = str(tmp_path / 'prostate')
    for d in range(1, 7):
        os.makedirs(os.path.join(base, 'Domain%d' % d, 'image'))
        os.makedirs(os.path.join(base, 'Domain%d' % d, 'mask'))
        for i in range(3):
            np.save(os.path.join(base, 'Domain%d' % d, 'image', '%d.npy' % i), np.full((32, 32, 3), d / 10.0, np.float32))
            np.save(os.path.join(base, 'Domain%d' % d, 'mask', '%d.npy' % i), rng.randint(0, 2, (32, 32)).astype(np.uint8))
    ds = Prostate_Multi(domain_idx_list=[2], base_dir=base, split='train', is_out_domain=True, test_domain_idx=0)
    np.random.seed(0); random.seed(0)
    doms = set()
    for k in range(30):
        img, other, lam, mask = ds[k % 3]
        assert img.shape == (32, 32, 3) and mask.dtype == torch.int64 and mask.shape == (32, 32)
        assert abs(float(img[0, 0, 0]) - 0.3) < 1e-6
        doms.add(int(round(float(other[0, 0, 0]) * 10)))
    assert doms == {2, 4, 5, 6}                                        # not the test domain (1), not its own (3)
