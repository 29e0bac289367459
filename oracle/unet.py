"""oracle.unet -- functional torch-CPU restatement of the reference U-Net.  TEST INFRASTRUCTURE ONLY.

Restates, from the reference checkout (code/networks/unet.py, code/networks/dsbn.py):
  ConvD.forward        unet.py:52-72     [maxpool2] conv3 bn conv3 bn act conv3 bn act   (bn1 has NO activation)
  ConvU.forward        unet.py:96-117    [conv3 bn act] up2 conv1 bn act cat[prev,y] conv3 bn act
  ConvU_Rec.forward    unet.py:139-165   conv3 dsbn act up2 conv1 dsbn act conv3 dsbn act
  Encoder              unet.py:248-271   Decoder unet.py:273-296   Rec_Decoder unet.py:299-322
  DomainSpecificBatchNorm2d.forward  dsbn.py:24-27  (picks bns[domain_label[0]])
  init loops           unet.py:257-262   kaiming_normal(fan_out) conv weights, BN w=1 b=0

The network is a pure function of a flat ``state`` dict whose keys/shapes/dtypes are exactly the
reference modules' ``state_dict()`` (so reference checkpoints load unchanged).  Parameters that need
gradients are plain leaf tensors with ``requires_grad``; backward is torch autograd.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

EPS = 1e-5
MOMENTUM = 0.1

# ----------------------------------------------------------------------------- rounding model (bf16 checker)
# The reference computes in fp32 throughout.  The HIP path's production mode stores activations, activation
# gradients and MFMA operands in bf16 (fp32 accumulation, fp32 BN statistics, fp32 parameters / optimizer).
# ``with rounding(torch.bfloat16):`` makes this restatement round at the SAME points (DESIGN.md section 2 'Numerics'):
#   forward   conv operands (activations after BN/act/pool/concat, weights) -> bf16; fp32 accumulate + bias;
#             BN batch statistics from the fp32 conv result; the stored (rounded) result is what gets normalised;
#             the 1x1 conv of ConvU/ConvU_Rec is evaluated BELOW the x2 upsample (exact commute in real arithmetic),
#             t and y = up2(t) are both stored rounded and BN2's statistics are taken from the stored y;
#             logits are stored rounded.
#   backward  the gradient w.r.t. every BN output (pre-activation) and w.r.t. every raw conv output is rounded
#             (they are what the dgrad / wgrad kernels read from HBM as bf16).
# Straight-through: rounding has gradient 1.  With ROUND = None (default) nothing below changes the fp32 restatement.
ROUND = None


class rounding:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global ROUND
        self.prev, ROUND = ROUND, self.dtype

    def __exit__(self, *exc):
        global ROUND
        ROUND = self.prev
        return False


def _rn(x):
    return x.to(ROUND).to(x.dtype)


def q(x):
    """Round values to the storage dtype (straight-through gradient)."""
    if ROUND is None:
        return x
    if not x.requires_grad:
        return _rn(x)
    return x + (_rn(x) - x).detach()


def qg(x):
    """Round the gradient that flows back into x."""
    if ROUND is not None and x.requires_grad:
        x.register_hook(_rn)
    return x


# ----------------------------------------------------------------------------- state construction
def _conv_entry(sd, name, cin, cout, k, gen):
    fan_out = cout * k * k
    fan_in = cin * k * k
    w = torch.randn(cout, cin, k, k, generator=gen) * math.sqrt(2.0 / fan_out)   # unet.py:259
    bound = 1.0 / math.sqrt(fan_in)                                               # nn.Conv2d default bias init
    b = (torch.rand(cout, generator=gen) * 2 - 1) * bound
    sd[name + '.weight'] = w
    sd[name + '.bias'] = b


def _bn_entry(sd, name, c, norm='bn'):
    """normalization(planes, norm) (unet.py:17-28): 'bn' nn.BatchNorm2d, 'gn' nn.GroupNorm(1, planes) (weight, bias only), 'in'
    nn.InstanceNorm2d(planes) (no state at all)."""
    if norm == 'in':
        return
    sd[name + '.weight'] = torch.ones(c)
    sd[name + '.bias'] = torch.zeros(c)
    if norm == 'gn':
        return
    sd[name + '.running_mean'] = torch.zeros(c)
    sd[name + '.running_var'] = torch.ones(c)
    sd[name + '.num_batches_tracked'] = torch.tensor(0, dtype=torch.long)


def _norm_entry(sd, name, c, num_domains):
    if num_domains is None:
        _bn_entry(sd, name, c)
    else:
        for d in range(num_domains):
            _bn_entry(sd, '%s.bns.%d' % (name, d), c)


def encoder_state(c=3, n=16, seed=0, norm='bn'):
    gen = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    chans = [c, n, 2 * n, 4 * n, 8 * n, 16 * n]
    for l in range(1, 6):
        p = 'convd%d' % l
        cin, co = chans[l - 1], chans[l]
        for j, ci in ((1, cin), (2, co), (3, co)):
            _conv_entry(sd, '%s.conv%d' % (p, j), ci, co, 3, gen)
            _bn_entry(sd, '%s.bn%d' % (p, j), co, norm)
    return sd


def decoder_state(n=16, num_classes=2, seed=1, norm='bn'):
    gen = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for l, planes, first in ((4, 16 * n, True), (3, 8 * n, False), (2, 4 * n, False), (1, 2 * n, False)):
        p = 'convu%d' % l
        if not first:
            _conv_entry(sd, p + '.conv1', 2 * planes, planes, 3, gen)
            _bn_entry(sd, p + '.bn1', planes, norm)
        _conv_entry(sd, p + '.conv2', planes, planes // 2, 1, gen)
        _bn_entry(sd, p + '.bn2', planes // 2, norm)
        _conv_entry(sd, p + '.conv3', planes, planes, 3, gen)
        _bn_entry(sd, p + '.bn3', planes, norm)
    _conv_entry(sd, 'out1', 2 * n, num_classes, 3, gen)
    return sd


def rec_decoder_state(n=16, num_classes=3, num_domains=3, seed=2):
    gen = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for l, planes in ((4, 16 * n), (3, 8 * n), (2, 4 * n), (1, 2 * n)):
        p = 'convu%d' % l
        h = planes // 2
        _conv_entry(sd, p + '.conv1', planes, h, 3, gen)
        _norm_entry(sd, p + '.bn1', h, num_domains)
        _conv_entry(sd, p + '.conv2', h, h, 1, gen)
        _norm_entry(sd, p + '.bn2', h, num_domains)
        _conv_entry(sd, p + '.conv3', h, h, 3, gen)
        _norm_entry(sd, p + '.bn3', h, num_domains)
    _conv_entry(sd, 'out1', n, num_classes, 3, gen)
    return sd


def is_param(key):
    return key.endswith('.weight') or key.endswith('.bias')


def param_keys(sd):
    return [k for k in sd if is_param(k)]


# ----------------------------------------------------------------------------- building blocks
def _act(x, slope):
    return F.relu(x) if slope == 0.0 else F.leaky_relu(x, slope)


def _conv(x, sd, name, pad):
    return F.conv2d(q(x), q(sd[name + '.weight']), sd[name + '.bias'], stride=1, padding=pad)


def _bn(x, sd, name, training, domain=None, stats_from_stored=False):
    """nn.BatchNorm2d (train: batch stats, biased var to normalise, unbiased into running_var,
    momentum 0.1) or DomainSpecificBatchNorm2d with ``domain`` = domain_label[0] (dsbn.py:26)."""
    if domain is not None:
        name = '%s.bns.%d' % (name, int(domain))
    if name + '.running_mean' not in sd:
        # what the state holds says which normalization() this is: weight + bias only = nn.GroupNorm(1, planes) (unet.py:20-21),
        # nothing = nn.InstanceNorm2d(planes) (unet.py:22-23); both normalise with the statistics of the input in train and eval mode
        assert ROUND is None, 'the bf16 rounding model covers bn / dsbn only'
        if name + '.weight' in sd:
            return F.group_norm(x, 1, sd[name + '.weight'], sd[name + '.bias'], EPS)
        return F.instance_norm(x, eps=EPS)
    if training:
        sd[name + '.num_batches_tracked'] += 1
    if ROUND is not None:
        return _bn_rounded(x, sd, name, training, stats_from_stored)
    return F.batch_norm(x, sd[name + '.running_mean'], sd[name + '.running_var'],
                        sd[name + '.weight'], sd[name + '.bias'], training, MOMENTUM, EPS)


def _bn_rounded(z, sd, name, training, stats_from_stored):
    """BatchNorm under the rounding model: statistics from the fp32 conv result (or from the stored tensor when
    the producer is the upsample), normalisation applied to the stored (rounded) tensor."""
    if stats_from_stored:
        zs = src = z                   # y = up2(t): already stored rounded; its gradient is never materialised (rd_up_bwd)
    else:
        qg(z)                          # dz = the dgrad / wgrad operand
        zs, src = q(z), z
    if training:
        mean = src.mean((0, 2, 3))
        var = src.var((0, 2, 3), unbiased=False)
        with torch.no_grad():
            n = src.numel() / src.shape[1]
            rm, rv = sd[name + '.running_mean'], sd[name + '.running_var']
            rm.mul_(1 - MOMENTUM).add_(MOMENTUM * mean)
            rv.mul_(1 - MOMENTUM).add_(MOMENTUM * var * (n / max(n - 1, 1)))
    else:
        mean, var = sd[name + '.running_mean'], sd[name + '.running_var']
    sc = sd[name + '.weight'] / torch.sqrt(var + EPS)
    y = zs * sc[None, :, None, None] + (sd[name + '.bias'] - mean * sc)[None, :, None, None]
    return qg(y)


def convd(x, sd, p, first, training, slope=0.0):
    if not first:
        x = F.max_pool2d(x, 2)
    x = _bn(_conv(x, sd, p + '.conv1', 1), sd, p + '.bn1', training)            # no activation (unet.py:59-60)
    y = _act(_bn(_conv(x, sd, p + '.conv2', 1), sd, p + '.bn2', training), slope)
    z = _act(_bn(_conv(y, sd, p + '.conv3', 1), sd, p + '.bn3', training), slope)
    return z


def _up_conv1(x, sd, p, training, domain, slope):
    """up x2 -> conv1x1 -> BN -> act (unet.py:104-108 / 152-156).  Rounding model: conv1x1 at low resolution, then
    the upsample (what the HIP path does; equal in real arithmetic because bilinear weights sum to 1)."""
    if ROUND is None:
        y = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
        return _act(_bn(_conv(y, sd, p + '.conv2', 0), sd, p + '.bn2', training, domain), slope)
    t = qg(q(_conv(x, sd, p + '.conv2', 0)))
    y = q(F.interpolate(t, scale_factor=2, mode='bilinear', align_corners=False))
    return _act(_bn(y, sd, p + '.bn2', training, domain, stats_from_stored=True), slope)


def convu(x, prev, sd, p, first, training, slope=0.0):
    if not first:
        x = _act(_bn(_conv(x, sd, p + '.conv1', 1), sd, p + '.bn1', training), slope)
    y = _up_conv1(x, sd, p, training, None, slope)
    y = torch.cat([prev, y], 1)                                                  # skip first (unet.py:110)
    y = _act(_bn(_conv(y, sd, p + '.conv3', 1), sd, p + '.bn3', training), slope)
    return y


def convu_rec(x, sd, p, domain, training, slope=0.0):
    x = _act(_bn(_conv(x, sd, p + '.conv1', 1), sd, p + '.bn1', training, domain), slope)
    y = _up_conv1(x, sd, p, training, domain, slope)
    y = _act(_bn(_conv(y, sd, p + '.conv3', 1), sd, p + '.bn3', training, domain), slope)
    return y


# ----------------------------------------------------------------------------- modules
def encoder_forward(x, sd, training=True, slope=0.0):
    feats = []
    for l in range(1, 6):
        x = convd(x, sd, 'convd%d' % l, l == 1, training, slope)
        feats.append(x)
    return feats


def decoder_forward(feats, sd, training=True, slope=0.0):
    y = convu(feats[-1], feats[-2], sd, 'convu4', True, training, slope)
    y = convu(y, feats[-3], sd, 'convu3', False, training, slope)
    y = convu(y, feats[-4], sd, 'convu2', False, training, slope)
    y = convu(y, feats[-5], sd, 'convu1', False, training, slope)
    return q(qg(_conv(y, sd, 'out1', 1)))


def rec_decoder_forward(x, sd, domain, training=True, slope=0.0):
    y = x
    for l in (4, 3, 2, 1):
        y = convu_rec(y, sd, 'convu%d' % l, domain, training, slope)
    return q(qg(_conv(y, sd, 'out1', 1)))


def clone_state(sd, requires_grad=False):
    out = OrderedDict()
    for k, v in sd.items():
        t = v.detach().clone()
        if requires_grad and is_param(k):
            t.requires_grad_(True)
        out[k] = t
    return out
