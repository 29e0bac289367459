"""oracle.losses -- torch-CPU restatement of the training losses.  TEST INFRASTRUCTURE ONLY.

Restates, from the reference checkout:
  dice_loss          code/utils/losses.py:8-16    one global ratio over ALL elements of (B,C,H,W)
  dice_loss_multi    code/utils/losses.py:18-33   per-class; ``i is ignore_index`` skips class ignore_index
  KD                 code/train.py:85-88          KLDivLoss()(p2.log(), p1) + KLDivLoss()(p1.log(), p2),
                                                  default reduction 'mean' == mean over every element
  BCELoss            torch.nn.BCELoss (train.py:202): mean of -(t*max(log p,-100) + (1-t)*max(log(1-p),-100))
"""
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
    loss = 0
    count = 0
    for i in range(num_classes):
        if i == ignore_index:            # reference uses ``is`` on small ints (losses.py:24)
            continue
        count += 1
        t_i = (target == i).float()
        intersect = torch.sum(score[:, i, ...] * t_i)
        y_sum = torch.sum(t_i * t_i)
        z_sum = torch.sum(score[:, i, ...] * score[:, i, ...])
        loss = loss + (1 - (2 * intersect + smooth) / (z_sum + y_sum + smooth))
    return loss / count


def bce(p, t):
    """nn.BCELoss(reduction='mean'): mean of -(t*max(log p,-100) + (1-t)*max(log(1-p),-100)); its
    backward is (p-t)/max(p*(1-p), 1e-12)/N (ATen binary_cross_entropy_backward), which stays finite
    at saturated p -- hence torch's own functional rather than a log/clamp graph."""
    return torch.nn.functional.binary_cross_entropy(p, t)


def kd(p_in, p_tgt):
    """KD(input, target), train.py:85-88.  KLDivLoss(reduction='mean')(x, y) = mean(y*(log y - x))
    with the xlogy convention 0*log 0 = 0; the symmetric sum equals mean((p1-p2)*(ln p1-ln p2))."""
    def kl(x_log, y):
        return torch.mean(torch.xlogy(y, y) - y * x_log)
    return kl(p_in.log(), p_tgt) + kl(p_tgt.log(), p_in)
