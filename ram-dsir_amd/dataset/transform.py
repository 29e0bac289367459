"""Host-side sample transforms used by the reference training pipeline (code/dataset/transform.py): Resize
(163-177), RandomScaleCrop (180-204) -> RandomCrop (16-44), Normalize (319-375), to_multilabel (10-14).  PIL on the
host, exactly like the reference; the arithmetic that dominates (RAM FFTs) runs on the GPU afterwards."""
import random

import numpy as np
import torch
from PIL import Image, ImageOps


def to_multilabel(pre_mask, classes=2):
    mask = np.zeros((pre_mask.shape[0], pre_mask.shape[1], classes))
    mask[pre_mask == 1] = [0, 1]
    mask[pre_mask == 2] = [1, 1]
    return mask


def _map_images(sample, f_img, f_mask):
    """Apply f_img to every image-like entry ('img', and 'img_freq' when a caller carries one along,
    transform.py:38-40) and f_mask to the mask; other keys are dropped like the reference does."""
    out = {'img': f_img(sample['img']), 'mask': f_mask(sample['mask'])}
    if 'img_freq' in sample:
        out['img_freq'] = f_img(sample['img_freq'])
    return out


class RandomCrop(object):
    """transform.py:16-44.  output_size = (w, h).  An image smaller than the crop is first padded on the right /
    bottom (image with 0, mask with 255 = background); the two offsets are ALWAYS drawn (x, then y) -- also when the
    image already has the crop size, where randint(0, 0) still advances python's generator -- so the draw order of a
    training run matches the reference's."""

    def __init__(self, output_size):
        self.output_size = output_size

    def __call__(self, sample):
        cw, ch = self.output_size[0], self.output_size[1]
        w, h = sample['img'].size
        pad = (0, 0, max(cw - w, 0), max(ch - h, 0))
        sample = _map_images(sample, lambda im: ImageOps.expand(im, border=pad, fill=0),
                             lambda m: ImageOps.expand(m, border=pad, fill=255))
        w, h = sample['img'].size
        x = random.randint(0, w - cw)
        y = random.randint(0, h - ch)
        box = (x, y, x + cw, y + ch)
        return _map_images(sample, lambda im: im.crop(box), lambda m: m.crop(box))


class Resize(object):
    """transform.py:163-177.  target_size = (w, h) as PIL orders it; bilinear image, nearest mask."""

    def __init__(self, target_size):
        self.target_size = target_size

    def __call__(self, sample):
        size = (self.target_size[0], self.target_size[1])
        return _map_images(sample, lambda im: im.resize(size, Image.BILINEAR), lambda m: m.resize(size, Image.NEAREST))


class RandomScaleCrop(object):
    """transform.py:180-204: with probability 1/2 enlarge width and height by independent factors U(1, 1.5)
    (draw order: the coin, the width factor, the height factor), then RandomCrop(size)."""

    def __init__(self, size):
        self.size = size
        self.crop = RandomCrop(self.size)

    def __call__(self, sample):
        img, mask = sample['img'], sample['mask']
        assert img.width == mask.width
        assert img.height == mask.height
        if random.random() > 0.5:
            w = int(random.uniform(1, 1.5) * img.size[0])
            h = int(random.uniform(1, 1.5) * img.size[1])
            sample = _map_images(sample, lambda im: im.resize((w, h), Image.BILINEAR), lambda m: m.resize((w, h), Image.NEAREST))
        return self.crop(sample)


class Normalize(object):
    """img -> float CHW in [-1,1]; gray mask -> 2-channel multilabel (transform.py:319-375)."""

    def __call__(self, sample):
        img = np.array(sample['img']).astype(np.float32).transpose((2, 0, 1))
        img /= 127.5
        img -= 1.0
        g = np.array(sample['mask']).astype(np.uint8)
        return {'img': torch.from_numpy(img).float(), 'mask': torch.from_numpy(fundus_mask(g)).float()}


def fundus_mask(gray_u8):
    """gray > 200 background, 51..200 disc only, <= 50 cup inside disc -> [cup, disc] (fundus.py:227-239)."""
    g = np.asarray(gray_u8).astype(np.uint8)
    t = np.zeros(g.shape)
    t[g > 200] = 255
    t[(g > 50) & (g < 201)] = 128
    lab = g.copy()
    lab[t == 0] = 2
    lab[t == 255] = 0
    lab[t == 128] = 1
    return to_multilabel(lab).transpose(2, 0, 1).astype(np.float32)
