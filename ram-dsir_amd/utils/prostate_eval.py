"""Prostate volume evaluation shared by train.py (per-epoch validation, code/train.py:134-192) and
test_prostate_volume.py (code/test_prostate_volume.py:79-161): per-volume min-max normalisation to [-1, 1], 2.5-D
stacks of three neighbouring slices, batches of `batch_size` frames, argmax of the softmax, suppression of slices
whose ground truth is empty, largest 3-D connected component, Dice (and HD95 / ASD for the test script)."""
import os

import numpy as np
import torch

from .metrics import asd, connectivity_region_analysis, dc, hd95
from .nifti import read_volume

DOMAIN_LIST = ['ISBI', 'ISBI_1.5', 'I2CVB', 'UCL', 'BIDMC', 'HK']          # code/train.py:77


def volume_files(data_dir, domain_name):
    """Image volumes of a domain: every file whose name has no 'segmentation' in it (train.py:140)."""
    return [f for f in os.listdir(os.path.join(data_dir, domain_name)) if 'segmentation' not in f]


def load_case(data_dir, domain_name, file_name):
    image = read_volume(os.path.join(data_dir, domain_name, file_name))
    mask = read_volume(os.path.join(data_dir, domain_name, file_name.replace('.nii.gz', '_segmentation.nii.gz')))
    return image, mask


def predict_volume(forward, image, mask, batch_size):
    """forward: (B,3,H,W) float32 tensor -> (B,K,H,W) logits.  Mirrors the reference loop exactly, including what
    reads like slips and therefore shapes the numbers: only floor(D / batch_size) batches run, so frames beyond
    that many batches are never predicted (test_prostate_volume.py:103); a batch is always `batch_size` wide, the
    slots beyond the frame list stay all-zero images and still take part in the batch statistics of a train-mode
    BatchNorm (:104-108); slices with an empty ground truth keep a zero prediction (:113-116)."""
    image = np.asarray(image)
    mx, mn = np.max(image), np.min(image)
    image = 2 * (image - mn) / (mx - mn) - 1
    mask = np.array(mask)
    mask[mask == 2] = 1
    pred_y = np.zeros(mask.shape)
    frame_list = list(range(1, image.shape[0] - 1))
    for ii in range(int(np.floor(image.shape[0] // batch_size))):
        vol = np.zeros([batch_size, 3, image.shape[1], image.shape[2]])
        frames = frame_list[ii * batch_size:(ii + 1) * batch_size]
        for idx, jj in enumerate(frames):
            vol[idx, ...] = image[jj - 1:jj + 2, ...].copy()
        logits = forward(torch.from_numpy(vol).float())
        pred = torch.max(torch.softmax(logits, dim=1), dim=1)[1].detach().cpu().numpy()
        for idx, jj in enumerate(frames):
            if np.sum(mask[jj, ...]) == 0:
                continue
            pred_y[jj, ...] = pred[idx, ...].copy()
    return connectivity_region_analysis(pred_y), mask


def evaluate_domain(forward, data_dir, domain_name, batch_size, with_surface=False, files=None):
    """Mean Dice (and mean HD95 / ASD) over the volumes of a domain."""
    files = volume_files(data_dir, domain_name) if files is None else files
    dice = hd = sd = 0.0
    n = 0
    for file_name in files:
        image, mask = load_case(data_dir, domain_name, file_name)
        post, mask = predict_volume(forward, image, mask, batch_size)
        dice += dc(post.astype(bool), mask.astype(bool))
        if with_surface:
            hd += hd95(post.astype(bool), mask.astype(bool))
            sd += asd(post.astype(bool), mask.astype(bool))
        n += 1
    n = max(n, 1)
    return dice / n, hd / n, sd / n
