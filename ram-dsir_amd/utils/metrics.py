"""Evaluation metrics of the reference that need no external package (code/utils/metrics.py:55-109,
code/utils/utils.py:19-28,45-96): Dice with +1 smoothing, largest-8-connected-component + hole filling
post-processing (skimage.label default connectivity == scipy structure of ones)."""
import numpy as np
import scipy.ndimage as ndi


def dice_coefficient_numpy(binary_segmentation, binary_gt_label):
    s = np.asarray(binary_segmentation, dtype=bool)
    g = np.asarray(binary_gt_label, dtype=bool)
    inter = float(np.logical_and(s, g).sum())
    return (2 * inter + 1.0) / (1.0 + float(s.sum()) + float(g.sum()))


def dice_coeff_2label(pred, target):
    target = np.asarray(target.cpu() if hasattr(target, 'cpu') else target)
    pred = np.asarray(pred)
    if pred.ndim == 3:
        return dice_coefficient_numpy(pred[0], target[0]), dice_coefficient_numpy(pred[1], target[1])
    cup = [dice_coefficient_numpy(pred[i, 0], target[i, 0]) for i in range(pred.shape[0])]
    disc = [dice_coefficient_numpy(pred[i, 1], target[i, 1]) for i in range(pred.shape[0])]
    return sum(cup) / len(cup), sum(disc) / len(disc)


def get_largest_fillhole(binary):
    binary = np.array(binary)
    lab, n = ndi.label(binary, structure=np.ones((3, 3)))
    if n:
        areas = ndi.sum(binary > 0, lab, index=np.arange(1, n + 1))
        binary[lab != (int(np.argmax(areas)) + 1)] = 0
    return ndi.binary_fill_holes(np.asarray(binary).astype(int))


def postprocessing(prediction, threshold=0.5, dataset='G'):
    p = np.asarray(prediction.cpu() if hasattr(prediction, 'cpu') else prediction)
    out = (p > threshold).astype(np.uint8)
    out[1] = get_largest_fillhole(out[1]).astype(np.uint8)
    out[0] = get_largest_fillhole(out[0]).astype(np.uint8)
    return out
