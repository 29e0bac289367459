"""Drop-in for the reference's code/networks/unet.py (Encoder / Decoder / Rec_Decoder and their blocks).

Same constructors, forward signatures, nn.Module behaviour and state_dict keys as the reference
(unet.py:248-322), so train.py / test_fundus_slice.py / test_prostate_volume.py and reference checkpoints
work unchanged.  The arithmetic is not PyTorch's: each module's forward is one fused launch list of HIP
kernels (ramdsir.engine), its backward another; tensors cross the boundary as NCHW fp32 like the reference.
The classes the reference never instantiates (Unet2D*, Discriminator) are out of scope.
"""
import torch
import torch.nn as nn

from networks.dsbn import DomainSpecificBatchNorm2d
from ramdsir import engine as E
from ramdsir import modules as M
from ramdsir import _lib as L


def count_params(model):
    param_num = sum(p.numel() for p in model.parameters())
    return param_num / 1e6


def normalization(planes, norm='gn', num_domains=None):
    if norm == 'bn':
        m = M.FusedBatchNorm2d(planes)
    elif norm == 'dsbn':
        m = DomainSpecificBatchNorm2d(planes, num_domains=num_domains)
    elif norm == 'gn':
        m = M.FusedGroupNorm(1, planes)
    elif norm == 'in':
        m = M.FusedInstanceNorm2d(planes)
    else:
        raise ValueError('Normalization type {} is not supporter'.format(norm))
    return m


def _plan_for(mod, key, N, training, build_graph, device):
    """Acquire (or build) the launch plan of a module called on its own."""
    def build():
        pl = E.Plan(mod._bank, M.storage_dtype(), N, M.group_starts(getattr(mod, '_norm', 'bn'), N), slope=mod._slope, training=training)
        build_graph(pl)
        pl.build(mod._wpack)
        pl.ws = E.workspace(pl.ws_bytes // 4, device)
        pl.bind_workspace(pl.ws)
        return pl
    return mod._acquire_plan(key + (training, M.storage_dtype()), build)


def _raw_input(pl, x, name):
    a = E.Act(pl, x.shape[0], x.shape[2], x.shape[3], x.shape[1], name=name)
    a.needs_grad = True
    return a


class ConvD(M.FusedModule):
    """unet.py:32-72: [MaxPool2d(2)] conv1 bn1 / conv2 bn2 act / conv3 bn3 act.  Inside Encoder the five blocks run as one
    launch list; called on its own the block builds its own (same kernels, the pool fused into conv1's read)."""
    _mname = 'convd'

    def __init__(self, inplanes, planes, norm='bn', first=False, activation='relu'):
        super(ConvD, self).__init__()
        self.first = first
        self.conv1 = M.FusedConv2d(inplanes, planes, 3, 1, 1, bias=True)
        self.bn1 = normalization(planes, norm)
        self.conv2 = M.FusedConv2d(planes, planes, 3, 1, 1, bias=True)
        self.bn2 = normalization(planes, norm)
        self.conv3 = M.FusedConv2d(planes, planes, 3, 1, 1, bias=True)
        self.bn3 = normalization(planes, norm)
        self._planes, self._norm = planes, norm
        self._finish_init(E.convd_specs(inplanes, planes, norm=norm), activation, init=False)

    def forward(self, x):
        _check_input(x, 'ConvD')
        N, Cc, H, W = x.shape
        if not self.first and (H % 2 or W % 2):
            raise ValueError('ConvD: H and W must be even for the 2x2 max-pool, got %dx%d' % (H, W))
        chunks = _image_chunks(self._norm, N)
        if chunks is not None:
            return torch.cat([self.forward(x[a:b]) for a, b in chunks], 0)
        self._ensure_bound(x.device)

        def graph(pl):
            pl.x_in = _raw_input(pl, x, 'input')
            pl.out = E.build_convd(pl, pl.x_in, L.SRC_RAW if self.first else L.SRC_POOL, self._planes, '', self._mname, self._norm)
            pl.out.g_written = True
        pl = _plan_for(self, (N, Cc, H, W), N, self._bn_training(), graph, x.device)
        return M.run_fused(self, pl, [pl.x_in], [pl.out], [x])[0]


class ConvU(M.FusedModule):
    """unet.py:75-117: [conv1 bn1 act] up2 conv2(1x1) bn2 act cat[prev, .] conv3 bn3 act."""
    _mname = 'convu'

    def __init__(self, planes, norm='bn', first=False, activation='relu'):
        super(ConvU, self).__init__()
        self.first = first
        if not self.first:
            self.conv1 = M.FusedConv2d(2 * planes, planes, 3, 1, 1, bias=True)
            self.bn1 = normalization(planes, norm)
        self.conv2 = M.FusedConv2d(planes, planes // 2, 1, 1, 0, bias=True)
        self.bn2 = normalization(planes // 2, norm)
        self.conv3 = M.FusedConv2d(planes, planes, 3, 1, 1, bias=True)
        self.bn3 = normalization(planes, norm)
        self._planes, self._norm = planes, norm
        self._finish_init(E.convu_specs(planes, first, norm=norm), activation, init=False)

    def forward(self, x, prev):
        _check_input(x, 'ConvU')
        _check_input(prev, 'ConvU')
        N = x.shape[0]
        chunks = _image_chunks(self._norm, N)
        if chunks is not None:
            return torch.cat([self.forward(x[a:b], prev[a:b]) for a, b in chunks], 0)
        self._ensure_bound(x.device)

        def graph(pl):
            pl.x_in, pl.prev_in = _raw_input(pl, x, 'x'), _raw_input(pl, prev, 'prev')
            pl.out = E.build_convu(pl, pl.x_in, pl.prev_in, self._planes, self.first, '', self._mname, self._norm)
            pl.out.g_written = True
        pl = _plan_for(self, (tuple(x.shape), tuple(prev.shape)), N, self._bn_training(), graph, x.device)
        return M.run_fused(self, pl, [pl.x_in, pl.prev_in], [pl.out], [x, prev])[0]


class ConvU_Rec(M.FusedModule):
    """unet.py:120-165: conv1 norm act up2 conv2(1x1) norm act conv3 norm act; norm='dsbn' picks bns[domain_label[0]]."""
    _mname = 'convu_rec'

    def __init__(self, planes, norm='bn', activation='relu', num_domains=None):
        super(ConvU_Rec, self).__init__()
        if norm in ('gn', 'in'):
            # the reference builds the restoration decoder with norm='dsbn' only (train.py:572,578); bn is kept for the block tests
            raise NotImplementedError('ConvU_Rec / Rec_Decoder: norm=%r has no HIP implementation (dsbn / bn)' % norm)
        self.conv1 = M.FusedConv2d(planes, planes // 2, 3, 1, 1, bias=True)
        self.bn1 = normalization(planes // 2, norm, num_domains)
        self.conv2 = M.FusedConv2d(planes // 2, planes // 2, 1, 1, 0, bias=True)
        self.bn2 = normalization(planes // 2, norm, num_domains)
        self.conv3 = M.FusedConv2d(planes // 2, planes // 2, 3, 1, 1, bias=True)
        self.bn3 = normalization(planes // 2, norm, num_domains)
        self._planes, self._dsbn = planes, norm == 'dsbn'
        self._finish_init(E.convu_rec_specs(planes, num_domains if self._dsbn else None), activation, init=False)

    def forward(self, x, domain_label=None):
        _check_input(x, 'ConvU_Rec')
        if self._dsbn and domain_label is None:
            raise TypeError('ConvU_Rec(norm="dsbn") needs domain_label (unet.py:142-146)')
        d = int(domain_label[0]) if domain_label is not None else None          # dsbn.py:26
        N = x.shape[0]
        self._ensure_bound(x.device)

        def graph(pl):
            pl.x_in = _raw_input(pl, x, 'x')
            pl.out = E.build_convu_rec(pl, pl.x_in, L.SRC_RAW, 0, -1, self._planes, [d] if self._dsbn else None, '', self._mname)
            pl.out.g_written = True
        pl = _plan_for(self, (tuple(x.shape), d), N, self._bn_training(), graph, x.device)
        return M.run_fused(self, pl, [pl.x_in], [pl.out], [x])[0]


def _image_chunks(norm, N):
    """gn / in keep one statistics group per image and a launch plan holds at most RD_MAX_GROUPS_C of them: a larger batch is run in
    chunks of that many images (exact: per-image statistics do not see the other images; the reference's nn.GroupNorm /
    nn.InstanceNorm2d have no batch limit).  None: the call fits one plan (always, for bn)."""
    if norm not in ('gn', 'in') or N <= L.MAXG:
        return None
    return [(i, min(i + L.MAXG, N)) for i in range(0, N, L.MAXG)]


def _check_input(x, name):
    if not (torch.is_tensor(x) and x.dim() == 4 and x.is_cuda):
        raise RuntimeError('%s: expected a 4-D CUDA tensor; the HIP path has no CPU fallback' % name)
    if x.shape[2] % 16 or x.shape[3] % 16:
        pass  # only the encoder needs H, W divisible by 16 (4 max-pools); checked there


class Encoder(M.FusedModule):
    _mname = 'enc'

    def __init__(self, c=3, n=16, norm='bn', activation='relu'):
        super(Encoder, self).__init__()
        self.convd1 = ConvD(c, n, norm, first=True, activation=activation)
        self.convd2 = ConvD(n, 2 * n, norm, activation=activation)
        self.convd3 = ConvD(2 * n, 4 * n, norm, activation=activation)
        self.convd4 = ConvD(4 * n, 8 * n, norm, activation=activation)
        self.convd5 = ConvD(8 * n, 16 * n, norm, activation=activation)
        self._c, self._n, self._norm = c, n, norm
        self._finish_init(E.encoder_specs(c, n, norm), activation)

    def forward(self, x):
        _check_input(x, 'Encoder')
        N, Cc, H, W = x.shape
        if H % 16 or W % 16:
            raise ValueError('Encoder: H and W must be multiples of 16 (four 2x2 max-pools), got %dx%d' % (H, W))
        chunks = _image_chunks(self._norm, N)
        if chunks is not None:
            parts = [self.forward(x[a:b]) for a, b in chunks]
            return [torch.cat([q[k] for q in parts], 0) for k in range(len(parts[0]))]
        self._ensure_bound(x.device)
        training = self._bn_training()

        def build():
            pl = E.Plan(self._bank, M.storage_dtype(), N, M.group_starts(self._norm, N), slope=self._slope, training=training)
            pl.x_in = E.Act(pl, N, H, W, Cc, name='input')
            pl.feats = E.build_encoder(pl, pl.x_in, n=self._n, mname=self._mname, norm=self._norm)
            for a in pl.feats:
                a.g_written = True                      # their gradient arrives from torch first (rd_grad_in)
            pl.build(self._wpack)
            pl.ws = E.workspace(pl.ws_bytes // 4, x.device)
            pl.bind_workspace(pl.ws)
            return pl
        pl = self._acquire_plan((N, H, W, training, M.storage_dtype()), build)
        return list(M.run_fused(self, pl, [pl.x_in], pl.feats, [x]))


class Decoder(M.FusedModule):
    _mname = 'dec'

    def __init__(self, n=16, num_classes=2, norm='bn', activation='relu'):
        super(Decoder, self).__init__()
        self.convu4 = ConvU(16 * n, norm, first=True, activation=activation)
        self.convu3 = ConvU(8 * n, norm, activation=activation)
        self.convu2 = ConvU(4 * n, norm, activation=activation)
        self.convu1 = ConvU(2 * n, norm, activation=activation)
        self.out1 = M.FusedConv2d(2 * n, num_classes, 3, padding=1)
        self._n, self._k, self._norm = n, num_classes, norm
        self._finish_init(E.decoder_specs(n, num_classes, norm), activation)

    def forward(self, feats):
        for f in feats:
            _check_input(f, 'Decoder')
        shapes = tuple(tuple(f.shape) for f in feats)
        N = feats[0].shape[0]
        chunks = _image_chunks(self._norm, N)
        if chunks is not None:
            return torch.cat([self.forward([f[a:b] for f in feats]) for a, b in chunks], 0)
        self._ensure_bound(feats[0].device)
        training = self._bn_training()

        def build():
            pl = E.Plan(self._bank, M.storage_dtype(), N, M.group_starts(self._norm, N), slope=self._slope, training=training)
            pl.ins = []
            for f in feats:
                a = E.Act(pl, N, f.shape[2], f.shape[3], f.shape[1], name='feat')
                a.needs_grad = True
                pl.ins.append(a)
            pl.logits = E.build_decoder(pl, pl.ins, n=self._n, num_classes=self._k, mname=self._mname, norm=self._norm)
            pl.logits.g_written = True
            pl.build(self._wpack)
            pl.ws = E.workspace(pl.ws_bytes // 4, feats[0].device)
            pl.bind_workspace(pl.ws)
            return pl
        pl = self._acquire_plan((shapes, training, M.storage_dtype()), build)
        return M.run_fused(self, pl, pl.ins, [pl.logits], list(feats))[0]


class Rec_Decoder(M.FusedModule):
    _mname = 'rec'

    def __init__(self, n=16, num_classes=2, norm='bn', activation='relu', num_domains=None):
        super(Rec_Decoder, self).__init__()
        self.convu4 = ConvU_Rec(16 * n, norm, activation=activation, num_domains=num_domains)
        self.convu3 = ConvU_Rec(8 * n, norm, activation=activation, num_domains=num_domains)
        self.convu2 = ConvU_Rec(4 * n, norm, activation=activation, num_domains=num_domains)
        self.convu1 = ConvU_Rec(2 * n, norm, activation=activation, num_domains=num_domains)
        self.out1 = M.FusedConv2d(n, num_classes, 3, padding=1)
        self._n, self._k = n, num_classes
        self._dsbn = (norm == 'dsbn')
        self._num_domains = num_domains
        self._finish_init(E.rec_decoder_specs(n, num_classes, num_domains if self._dsbn else None), activation)

    def forward(self, x, domain_label=None):
        _check_input(x, 'Rec_Decoder')
        if self._dsbn and domain_label is None:
            raise TypeError('Rec_Decoder(norm="dsbn") needs domain_label (unet.py:142-146)')
        d = int(domain_label[0]) if domain_label is not None else None         # dsbn.py:26
        N, Cc, H, W = x.shape
        self._ensure_bound(x.device)
        training = self._bn_training()

        def build():
            pl = E.Plan(self._bank, M.storage_dtype(), N, [0, N], slope=self._slope, training=training)
            pl.x_in = E.Act(pl, N, H, W, Cc, name='bottleneck')
            pl.x_in.needs_grad = True
            pl.out = E.build_rec_decoder(pl, pl.x_in, 0, -1, [d] if self._dsbn else None, n=self._n, num_classes=self._k,
                                         mname=self._mname)
            pl.out.g_written = True
            pl.build(self._wpack)
            pl.ws = E.workspace(pl.ws_bytes // 4, x.device)
            pl.bind_workspace(pl.ws)
            return pl
        pl = self._acquire_plan((N, Cc, H, W, d, training, M.storage_dtype()), build)
        return M.run_fused(self, pl, [pl.x_in], [pl.out], [x])[0]
