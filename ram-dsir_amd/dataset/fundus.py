"""Drop-in for the reference's code/dataset/fundus.py.

Same constructors and on-disk format (``<base>/DomainK_{train,test}.list`` with "<img> <mask>" lines and the
partner lists ``<base>/DomainK/train.list``, fundus.py:143,206).  One deliberate change: the
reference runs the RAM FFTs on the CPU inside ``__getitem__`` in DataLoader workers; workers must not touch
the GPU, so here ``Fundus_Multi.__getitem__`` returns the PIECES RAM needs --
``(img_hwc uint8, partner_hwc uint8, lam, mask)`` (the decoded pixels; the GPU kernel reads 1 byte per value) with the reference's sampling semantics
(partner domain != own domain if is_out_domain, never the test domain; partner resized to 256x256 BILINEAR;
lam = random.randint(1,10)/10) -- and the FFTs run on the GPU for the whole batch (ramdsir.ram / the fused step).
``ram_collate`` turns a list of such samples into what the reference's loader yields: (img, img_freq, mask).
"""
import os
import random

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

from dataset.transform import fundus_mask, to_multilabel     # noqa: F401
from ramdsir import ram as _ram

DOMAINS = ['Domain1', 'Domain2', 'Domain3', 'Domain4']


def extract_amp_spectrum(img_np):
    """fundus.py:13-19: amplitude of the 2-D FFT over the last two axes of a CHW array; numpy float32 array, computed by
    the HIP kernels (rd_ram_amp).  Main process only: DataLoader workers must not touch the GPU (the training path
    mixes whole batches with rd_ram_mix instead)."""
    return _ram.extract_amp_spectrum_gpu(np.asarray(img_np, dtype=np.float32)).cpu().numpy()


def low_freq_mutate_np(amp_src, amp_trg, L=0.1):
    """fundus.py:21-39; like the reference the ratio is drawn HERE: random.randint(1, 10) / 10."""
    ratio = random.randint(1, 10) / 10
    return _ram.low_freq_mutate_gpu(amp_src, amp_trg, ratio, L).cpu().numpy()


def source_to_target_freq(src_img, amp_trg, L=0.1, lam=None):
    """fundus.py:41-61 for one HWC image and the partner's amplitude spectrum (CHW array from extract_amp_spectrum); the
    ratio is drawn like the reference does inside low_freq_mutate_np unless `lam` is given.  HWC numpy float32."""
    if lam is None:
        lam = random.randint(1, 10) / 10
    return _ram.source_to_target_freq_gpu(np.asarray(src_img, dtype=np.float32), amp_trg, lam, L).cpu().numpy()


def _read_list(path):
    with open(path, 'r') as f:
        return [l.replace('\n', '') for l in f.readlines()]


class Fundus(Dataset):
    """Test dataset (fundus.py:64-125): returns (img, mask, mask_orig, id) after the transform."""

    def __init__(self, domain_idx=None, base_dir=None, split='train', num=None, transform=None, is_ra=False):
        self.transform, self.base_dir, self.split = transform, base_dir, split
        self.id_path = _read_list(os.path.join(base_dir, '%s_%s.list' % (DOMAINS[domain_idx], 'train' if split == 'train' else 'test')))
        if num is not None:
            self.id_path = self.id_path[:num]
        print('total {} samples'.format(len(self.id_path)))

    def __len__(self):
        return len(self.id_path)

    def __getitem__(self, index):
        id = self.id_path[index]
        img = Image.open(os.path.join(self.base_dir, id.split(' ')[0]))
        mask = Image.open(os.path.join(self.base_dir, id.split(' ')[1])).convert('L')
        sample = {'img': img, 'mask': mask}
        mask_orig = torch.from_numpy(fundus_mask(np.array(mask))).float()
        if self.transform:
            sample = self.transform(sample)
        return sample['img'], sample['mask'], mask_orig, id


class Fundus_Multi(Dataset):
    def __init__(self, domain_idx_list=None, base_dir=None, split='train', num=None, transform=None, is_freq=True,
                 is_out_domain=False, test_domain_idx=None):
        self.transform, self.base_dir, self.num = transform, base_dir, num
        self.domain_name = list(DOMAINS)
        self.domain_idx_list, self.split, self.is_freq = domain_idx_list, split, is_freq
        self.is_out_domain, self.test_domain_idx = is_out_domain, test_domain_idx
        self.id_path = []
        for d in domain_idx_list:
            self.id_path += _read_list(os.path.join(base_dir, '%s_%s.list' % (self.domain_name[d], 'train' if split == 'train' else 'test')))
        if num is not None:
            self.id_path = self.id_path[:num]
        self._partner_lists = {}
        print('total {} samples'.format(len(self.id_path)))

    def __len__(self):
        return len(self.id_path)

    def _partner(self, cur_domain_name):
        train_domain_name = self.domain_name.copy()
        train_domain_name.remove(self.domain_name[self.test_domain_idx])
        domain_list = train_domain_name.copy()
        if self.is_out_domain:
            domain_list.remove(cur_domain_name)
        other = np.random.choice(domain_list, 1)[0]                       # fundus.py:205
        if other not in self._partner_lists:
            self._partner_lists[other] = _read_list(os.path.join(self.base_dir, other, 'train.list'))
        other_id = np.random.choice(self._partner_lists[other]).split(' ')[0]          # fundus.py:208
        img = Image.open(os.path.join(self.base_dir, other, other_id)).resize((256, 256), Image.BILINEAR)
        return np.array(img)                                              # uint8 HWC (the reference converts to float32: same values)

    def __getitem__(self, index):
        id = self.id_path[index]
        img = Image.open(os.path.join(self.base_dir, id.split(' ')[0]))
        mask = Image.open(os.path.join(self.base_dir, id.split(' ')[1])).convert('L')
        cur_domain_name = id.split(' ')[0].split('/')[0]
        sample = {'img': img, 'mask': mask}
        if self.transform:
            sample = self.transform(sample)
        img = np.array(sample['img'])                                     # HWC uint8, 0..255: 1 byte per value to the GPU
        mask = torch.from_numpy(fundus_mask(np.array(sample['mask']))).float()
        if not self.is_freq:
            return torch.from_numpy(img.astype(np.float32).transpose(2, 0, 1) / 127.5 - 1.0).float(), mask
        other = self._partner(cur_domain_name)
        if other.shape != img.shape:                                      # reference assumes 256x256 crops (fundus.py:209)
            other = np.array(Image.fromarray(other).resize((img.shape[1], img.shape[0]), Image.BILINEAR))
        lam = random.randint(1, 10) / 10                                  # fundus.py:35
        return torch.from_numpy(img), torch.from_numpy(other), torch.tensor(lam, dtype=torch.float32), mask


def ram_collate(src, trg, lam, dataset='fundus', dtype=torch.float32):
    """(img, img_freq) NCHW fp32 in [-1,1] for a batch of RAM pieces already on the GPU."""
    return _ram.source_to_target_freq_batch(src, trg, lam, dataset, dtype)
