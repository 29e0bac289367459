"""Drop-in for the two loss functions the reference imports (code/utils/losses.py:8-33).  The fused trainer does
not call these (its losses are HIP kernels, rd_seg_loss / rd_rec_loss); they exist for code written against
the reference API and operate on whatever device their tensors live on."""
import torch


def dice_loss(score, target):
    target = target.float()
    smooth = 1e-5
    intersect = torch.sum(score * target)
    y_sum = torch.sum(target * target)
    z_sum = torch.sum(score * score)
    return 1 - (2 * intersect + smooth) / (z_sum + y_sum + smooth)


def dice_loss_multi(score, target, num_classes, ignore_index=255):
    target = target.float()
    smooth = 1e-5
    loss, count = 0, 0
    for i in range(num_classes):
        if i == ignore_index:
            continue
        count += 1
        t = (target == i).float()
        intersect = torch.sum(score[:, i, ...] * t)
        loss = loss + (1 - (2 * intersect + smooth) / (torch.sum(score[:, i, ...] ** 2) + torch.sum(t * t) + smooth))
    return loss / count
