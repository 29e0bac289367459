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


def dice(input, target, ignore_index=None):
    """metrics.py:28-38 (torch tensors, +1 smoothing over the flattened batch); unused by the scripts, kept for API parity."""
    smooth = 1.
    iflat, tflat = input.clone().view(-1), target.clone().view(-1)
    if ignore_index is not None:
        m = tflat == ignore_index
        tflat[m] = 0
        iflat[m] = 0
    return (2. * (iflat * tflat).sum() + smooth) / (iflat.sum() + tflat.sum() + smooth)


def dice_multi(input, target, num_classes=3, ignore_index=None):
    """metrics.py:40-53: mean over classes (minus ignore_index) of the label-map Dice, 1e-5 smoothing."""
    smooth, count, total = 1e-5, 0, 0
    for i in range(num_classes):
        if i == ignore_index:
            continue
        count += 1
        a, b = (input == i), (target == i)
        total = total + (2 * (a * b).sum() + smooth) / (a.sum() + b.sum() + smooth)
    return total / count


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


def postprocess_binary(mask_u8):
    """postprocessing() after the threshold: largest 8-connected component + hole filling per channel, disc (1) then cup (0)
    like utils/utils.py:81-89.  mask_u8: (2, H, W) uint8 0/1."""
    out = np.array(mask_u8, dtype=np.uint8)
    out[1] = get_largest_fillhole(out[1]).astype(np.uint8)
    out[0] = get_largest_fillhole(out[0]).astype(np.uint8)
    return out


def post_and_dice(args):
    """(thresholded prediction (2,H,W) uint8, target (2,H,W) uint8) -> (cup dice, disc dice): one validation image
    (train.py:116-118); a top-level function so that a process pool can run the images of a batch in parallel."""
    mask_u8, target_u8 = args
    post = postprocess_binary(mask_u8)
    return dice_coefficient_numpy(post[0], target_u8[0]), dice_coefficient_numpy(post[1], target_u8[1])


# ---------------------------------------------------------------------------------------------------------------
# medpy.metric.binary restated.  The reference imports `medpy` (test_fundus_slice.py:19,125-136,
# test_prostate_volume.py:14,121-126, utils/metrics.py:2,20; no version pinned, the package is not vendored and not
# installed here).  These follow medpy's published definitions (medpy/metric/binary.py, 0.4.x): `dc` = 2|A&B| /
# (|A|+|B|), 0 when both are empty; surface distances = Euclidean distance transform of the complement of the
# REFERENCE's border sampled on the RESULT's border, a border being `x XOR binary_erosion(x)` with the connectivity-1
# (face) structuring element; `hd95` = 95th percentile of the two directed distance sets stacked; `asd` = mean of the
# directed set result -> reference.  Empty operands raise RuntimeError as medpy does.
def _as_bool(a):
    return np.atleast_1d(np.asarray(a).astype(bool))


def dc(result, reference):
    result, reference = _as_bool(result), _as_bool(reference)
    inter = np.count_nonzero(result & reference)
    a, b = np.count_nonzero(result), np.count_nonzero(reference)
    try:
        return 2.0 * inter / float(a + b)
    except ZeroDivisionError:
        return 0.0


def _surface_distances(result, reference, voxelspacing=None, connectivity=1):
    result, reference = _as_bool(result), _as_bool(reference)
    if voxelspacing is not None:
        voxelspacing = np.asarray(voxelspacing, dtype=np.float64)
        if voxelspacing.ndim == 0:
            voxelspacing = np.repeat(voxelspacing, result.ndim)
    footprint = ndi.generate_binary_structure(result.ndim, connectivity)
    if np.count_nonzero(result) == 0:
        raise RuntimeError('The first supplied array does not contain any binary object.')
    if np.count_nonzero(reference) == 0:
        raise RuntimeError('The second supplied array does not contain any binary object.')
    result_border = result ^ ndi.binary_erosion(result, structure=footprint, iterations=1)
    reference_border = reference ^ ndi.binary_erosion(reference, structure=footprint, iterations=1)
    dt = ndi.distance_transform_edt(~reference_border, sampling=voxelspacing)
    return dt[result_border]


def hd95(result, reference, voxelspacing=None, connectivity=1):
    hd1 = _surface_distances(result, reference, voxelspacing, connectivity)
    hd2 = _surface_distances(reference, result, voxelspacing, connectivity)
    return float(np.percentile(np.hstack((hd1, hd2)), 95))


def asd(result, reference, voxelspacing=None, connectivity=1):
    return float(_surface_distances(result, reference, voxelspacing, connectivity).mean())


def connectivity_region_analysis(mask):
    """code/utils/utils.py:30-42 `_connectivity_region_analysis`: keep the largest connected component of a volume
    (scipy.ndimage.label with its default face connectivity; sizes[0] is the background's sum, i.e. 0 for a
    binary mask).  Returns an integer array like the reference."""
    mask = np.asarray(mask)
    label_im, nb = ndi.label(mask)
    sizes = ndi.sum(mask, label_im, range(nb + 1))
    keep = int(np.argmax(sizes))
    out = np.zeros_like(label_im)
    out[label_im == keep] = 1
    if keep == 0:                      # the reference's two in-place assignments: label 0 stays 0 first, then 0 == argmax -> all ones
        out = np.ones_like(label_im)
    return out
