"""oracle.masks -- numpy restatement of the Fundus mask encoding (M1).  TEST INFRASTRUCTURE ONLY.

Restates code/dataset/fundus.py:227-239 + code/dataset/transform.py:10-14:
  gray > 200        -> background  [cup, disc] = [0, 0]
  50 < gray <= 200  -> disc only                 = [0, 1]
  gray <= 50        -> cup (inside the disc)     = [1, 1]
"""
import numpy as np


def fundus_mask_multilabel(gray_u8):
    g = np.asarray(gray_u8).astype(np.uint8)
    tmp = np.zeros(g.shape)
    tmp[g > 200] = 255
    tmp[(g > 50) & (g < 201)] = 128
    lab = g.copy()
    lab[tmp == 0] = 2
    lab[tmp == 255] = 0
    lab[tmp == 128] = 1
    out = np.zeros((g.shape[0], g.shape[1], 2))
    out[lab == 1] = [0, 1]
    out[lab == 2] = [1, 1]
    return out.transpose(2, 0, 1).astype(np.float32)
