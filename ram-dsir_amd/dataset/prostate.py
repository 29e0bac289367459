"""Drop-in for the reference's code/dataset/prostate.py: ``<base>/DomainK/{image,mask}/*.npy`` slices of shape
(S,S,3) in [-1,1] and (S,S) labels.  As in dataset/fundus.py, the train ``__getitem__`` returns the RAM pieces
(img_hwc, partner_hwc, lam, mask) and the FFTs (clip to [-1,1], prostate.py:188) run on the GPU per batch."""
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from dataset.fundus import extract_amp_spectrum, low_freq_mutate_np, source_to_target_freq     # noqa: F401  (prostate.py:10-62 is the same trio)

DOMAINS = ['Domain1', 'Domain2', 'Domain3', 'Domain4', 'Domain5', 'Domain6']


class Prostate(Dataset):
    def __init__(self, domain_idx=None, base_dir=None, split='train', num=None, transform=None):
        self.base_dir, self.split = base_dir, split
        lst = os.listdir(os.path.join(base_dir, DOMAINS[domain_idx], 'image'))
        self.id_path = [DOMAINS[domain_idx] + '/image/' + i for i in lst]
        if num is not None:
            self.id_path = self.id_path[:num]
        print('total {} samples'.format(len(self.id_path)))

    def __len__(self):
        return len(self.id_path)

    def __getitem__(self, index):
        id = self.id_path[index]
        img = np.load(os.path.join(self.base_dir, id))
        mask = np.load(os.path.join(self.base_dir, id.replace('image', 'mask')))
        return torch.from_numpy(img.transpose(2, 0, 1)).float(), torch.from_numpy(mask).long(), id.split('/')[-1]


class Prostate_Multi(Dataset):
    def __init__(self, domain_idx_list=None, base_dir=None, split='train', num=None, transform=None, is_freq=True,
                 is_out_domain=False, test_domain_idx=None):
        self.base_dir, self.num, self.domain_name = base_dir, num, list(DOMAINS)
        self.domain_idx_list, self.split, self.is_freq = domain_idx_list, split, is_freq
        self.is_out_domain, self.test_domain_idx = is_out_domain, test_domain_idx
        self.id_path = []
        for d in domain_idx_list:
            lst = os.listdir(os.path.join(base_dir, self.domain_name[d], 'image'))
            self.id_path += [self.domain_name[d] + '/image/' + i for i in lst]
        if num is not None:
            self.id_path = self.id_path[:num]
        print('total {} samples'.format(len(self.id_path)))

    def __len__(self):
        return len(self.id_path)

    def __getitem__(self, index):
        train_domain_name = self.domain_name.copy()
        train_domain_name.remove(self.domain_name[self.test_domain_idx])
        id = self.id_path[index]
        img = np.load(os.path.join(self.base_dir, id)).astype(np.float32)
        mask = torch.from_numpy(np.load(os.path.join(self.base_dir, id.replace('image', 'mask')))).long()
        if self.split == 'test' or not self.is_freq:
            return torch.from_numpy(img.transpose(2, 0, 1)).float(), mask
        domain_list = train_domain_name.copy()
        if self.is_out_domain:
            domain_list.remove(id.split('/')[0])
        other_domain_name = np.random.choice(domain_list, 1)[0]                          # prostate.py:182
        other_id = np.random.choice(os.listdir(os.path.join(self.base_dir, other_domain_name, 'image')))
        other = np.load(os.path.join(self.base_dir, other_domain_name, 'image', other_id)).astype(np.float32)
        lam = random.randint(1, 10) / 10
        return torch.from_numpy(img), torch.from_numpy(other), torch.tensor(lam, dtype=torch.float32), mask
