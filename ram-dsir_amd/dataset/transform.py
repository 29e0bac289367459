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


class RandomCrop(object):
    def __init__(self, size, padding=0):
        self.size = (int(size), int(size)) if isinstance(size, (int, float)) else size   # (h, w)
        self.padding = padding

    def __call__(self, sample):
        img, mask = sample['img'], sample['mask']
        if self.padding > 0:
            img = ImageOps.expand(img, border=self.padding, fill=0)
            mask = ImageOps.expand(mask, border=self.padding, fill=0)
        assert img.width == mask.width and img.height == mask.height
        w, h = img.size
        th, tw = self.size
        if w == tw and h == th:
            return {'img': img, 'mask': mask}
        if w < tw or h < th:
            img = img.resize((tw, th), Image.BILINEAR)
            mask = mask.resize((tw, th), Image.NEAREST)
            return {'img': img, 'mask': mask}
        x1 = random.randint(0, w - tw)
        y1 = random.randint(0, h - th)
        return {'img': img.crop((x1, y1, x1 + tw, y1 + th)), 'mask': mask.crop((x1, y1, x1 + tw, y1 + th))}


class Resize(object):
    def __init__(self, size):
        self.size = tuple(reversed(size))                    # size: (h, w)

    def __call__(self, sample):
        img, mask = sample['img'], sample['mask']
        assert img.width == mask.width and img.height == mask.height
        return {'img': img.resize(self.size, Image.BILINEAR), 'mask': mask.resize(self.size, Image.NEAREST)}


class RandomScaleCrop(object):
    def __init__(self, size):
        self.size = size
        self.crop = RandomCrop(self.size)

    def __call__(self, sample):
        img, mask = sample['img'], sample['mask']
        assert img.width == mask.width and img.height == mask.height
        seed = random.random()
        if seed > 0.5:
            w = int(random.uniform(1, 1.5) * img.size[0])
            h = int(random.uniform(1, 1.5) * img.size[1])
            img, mask = img.resize((w, h), Image.BILINEAR), mask.resize((w, h), Image.NEAREST)
            sample['img'], sample['mask'] = img, mask
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
