"""Deterministic synthetic dataset trees in the reference's on-disk formats (TEST INFRASTRUCTURE).

Fundus (code/dataset/fundus.py:140-150,206): ``<base>/DomainK_{train,test}.list`` with "<img> <mask>" lines relative to
``<base>``, the partner lists ``<base>/DomainK/train.list`` relative to ``<base>/DomainK``, RGB PNG ROIs and gray PNG
masks (0 = cup, 128 = disc rim, 255 = background).  Prostate (code/dataset/prostate.py:132-137,183-185):
``<base>/DomainK/{image,mask}/*.npy`` with (S,S,3) float32 slices in [-1,1] and (S,S) uint8 labels.

Used by tests/golden/make_golden.py (feeds the REFERENCE's datasets in the build container) and by the tests (feed
the drop-in datasets); PNG / npy round trips are lossless, so both sides see identical pixels.
"""
import os

import numpy as np
from PIL import Image


def _smooth_rgb(rng, h, w):
    """uint8 image with large-scale structure + pixel noise (so that resampling and RAM are non-trivial)."""
    low = rng.uniform(0, 255, (max(h // 16, 2), max(w // 16, 2), 3)).astype(np.float32)
    img = np.stack([np.array(Image.fromarray(low[..., c]).resize((w, h), Image.BILINEAR)) for c in range(3)], -1)
    img = 0.8 * img + 0.2 * rng.uniform(0, 255, (h, w, 3))
    return np.clip(np.round(img), 0, 255).astype(np.uint8)


def _disc_mask(rng, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    cy, cx = rng.uniform(0.4 * h, 0.6 * h), rng.uniform(0.4 * w, 0.6 * w)
    r = rng.uniform(0.2, 0.35) * min(h, w)
    d2 = (yy - cy) ** 2 + (xx - cx) ** 2
    m = np.full((h, w), 255, np.uint8)
    m[d2 < r * r] = 128
    m[d2 < (0.5 * r) ** 2] = 0
    return m


def make_fundus_tree(root, n_train=4, n_test=2, hw=(280, 300), seed=20221, vary=True):
    """vary=True: every file gets its own size (h + 8 d, w - 4 i) -- exercises Resize; vary=False: all hw (the real ROIs are all
    800x800, and the test dataset's native-size masks must collate)."""
    rng = np.random.RandomState(seed)
    base = os.path.join(root, 'fundus')
    h, w = hw
    for d in range(1, 5):
        for split, n in (('train', n_train), ('test', n_test)):
            for sub in ('image', 'mask'):
                os.makedirs(os.path.join(base, 'Domain%d' % d, split, 'ROIs', sub), exist_ok=True)
            lines, partner = [], []
            for i in range(n):
                ri = 'Domain%d/%s/ROIs/image/d%d_%s_%02d.png' % (d, split, d, split, i)
                rm = ri.replace('/image/', '/mask/')
                hh, ww = (h + 8 * d, w - 4 * i) if vary else (h, w)
                Image.fromarray(_smooth_rgb(rng, hh, ww)).save(os.path.join(base, ri))
                Image.fromarray(_disc_mask(rng, hh, ww)).save(os.path.join(base, rm))
                lines.append(ri + ' ' + rm)
                partner.append(ri.split('/', 1)[1] + ' ' + rm.split('/', 1)[1])
            with open(os.path.join(base, 'Domain%d_%s.list' % (d, split)), 'w') as f:
                f.write('\n'.join(lines) + '\n')
            if split == 'train':
                with open(os.path.join(base, 'Domain%d' % d, 'train.list'), 'w') as f:
                    f.write('\n'.join(partner) + '\n')
    return base


def make_prostate_tree(root, n=4, S=64, seed=20222):
    rng = np.random.RandomState(seed)
    base = os.path.join(root, 'prostate')
    for d in range(1, 7):
        for sub in ('image', 'mask'):
            os.makedirs(os.path.join(base, 'Domain%d' % d, sub), exist_ok=True)
        for i in range(n):
            img = _smooth_rgb(rng, S, S).astype(np.float32) / 127.5 - 1.0
            msk = (_disc_mask(rng, S, S) < 200).astype(np.uint8)
            np.save(os.path.join(base, 'Domain%d' % d, 'image', 'd%d_s%02d.npy' % (d, i)), img.astype(np.float32))
            np.save(os.path.join(base, 'Domain%d' % d, 'mask', 'd%d_s%02d.npy' % (d, i)), msk)
    return base


class DrawLog:
    """Records every draw the datasets make from python's ``random`` and ``numpy.random`` (the two generators the
    reference uses: transform.py:22-41,186-194, fundus.py:35,205,208), so that the sampling ORDER and VALUES of the
    drop-in can be compared with the reference's."""

    def __init__(self):
        self.log = []

    def __enter__(self):
        import random
        self._r = (random.random, random.uniform, random.randint)
        self._c = np.random.choice
        log = self.log

        def w_random():
            v = self._r[0]()
            log.append('random:%.17g' % v)
            return v

        def w_uniform(a, b):
            v = self._r[1](a, b)
            log.append('uniform:%.17g' % v)
            return v

        def w_randint(a, b):
            v = self._r[2](a, b)
            log.append('randint(%d,%d):%d' % (a, b, v))
            return v

        def w_choice(a, *args, **kw):
            v = self._c(a, *args, **kw)
            log.append('choice:%s' % (str(np.asarray(v).reshape(-1)[0]).strip()))
            return v
        random.random, random.uniform, random.randint = w_random, w_uniform, w_randint
        np.random.choice = w_choice
        return self

    def __exit__(self, *exc):
        import random
        random.random, random.uniform, random.randint = self._r
        np.random.choice = self._c
        return False
