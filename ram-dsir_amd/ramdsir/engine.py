"""Host-side execution engine: turns the RAM-DSIR network (code/networks/unet.py) into a static list of
HIP launches over preallocated HBM buffers.

Design (MI355X-first):
  * activations / gradients are NHWC in `dtype` (bf16 or fp32); every conv output is stored RAW
    (pre-BatchNorm); the BN affine + activation + max-pool / bilinear x2 / concat of the reference graph
    live in the tile loaders of the consuming conv ("normalise on read") -- no intermediate tensor of
    the reference between two convs is ever materialised;
  * images that share BatchNorm statistics form a *group*; the two encoder/seg-decoder passes of a
    training step run as ONE batch of 2B images with 2 groups, the per-domain restoration-decoder calls
    as ONE batch with one group per domain (DomainSpecificBatchNorm2d);
  * the 1x1 conv of ConvU / ConvU_Rec is evaluated below the upsample (it commutes with bilinear
    interpolation); BN2 statistics are still those of the upsampled tensor (rd_up_stats);
  * backward: dgrad epilogues write the gradient w.r.t. the producer's BN output (activation mask,
    max-pool scatter, skip accumulation) and the two BN-backward sums; BN backward itself is the
    per-channel P*g+Q*z+R folded into the reads of the next dgrad / wgrad.
"""
import ctypes as C
import os

import torch

from . import _lib as L
from . import tuning as T

EPS = 1e-5
MOMENTUM = 0.1


def workspace(n, device):
    """Scratch float32 buffer whose contents the kernels must never rely on.  RD_POISON=1 fills it with NaN, so a kernel
    that reads a workspace element nobody wrote shows up as NaN in every GPU test instead of as a rare flake."""
    if os.environ.get('RD_POISON') == '1':
        return torch.full((max(int(n), 1),), float('nan'), dtype=torch.float32, device=device)
    return torch.empty(max(int(n), 1), dtype=torch.float32, device=device)


# ------------------------------------------------------------------------------------------------ specs
def _conv_spec(out, name, cin, cout, k):
    out.append((name + '.weight', (cout, cin, k, k), 'param', torch.float32))
    out.append((name + '.bias', (cout,), 'param', torch.float32))


def _bn_spec(out, name, c, norm='bn'):
    """state_dict entries of one normalization(planes, norm) site (unet.py:17-28): nn.BatchNorm2d has weight, bias and three buffers,
    nn.GroupNorm(1, planes) weight and bias, nn.InstanceNorm2d(planes) (affine=False, no running statistics) nothing."""
    if norm == 'in':
        return
    out.append((name + '.weight', (c,), 'param', torch.float32))
    out.append((name + '.bias', (c,), 'param', torch.float32))
    if norm == 'gn':
        return
    out.append((name + '.running_mean', (c,), 'buffer', torch.float32))
    out.append((name + '.running_var', (c,), 'buffer', torch.float32))
    out.append((name + '.num_batches_tracked', (), 'buffer', torch.int64))


def _norm_spec(out, name, c, num_domains):
    if num_domains is None:
        _bn_spec(out, name, c)
    else:
        for d in range(num_domains):
            _bn_spec(out, '%s.bns.%d' % (name, d), c)


def convd_specs(cin, cout, prefix='', out=None, norm='bn'):
    """networks.unet.ConvD (unet.py:32-50): conv1,bn1,conv2,bn2,conv3,bn3.  prefix '' = the block on its own."""
    out = [] if out is None else out
    for j, ci in ((1, cin), (2, cout), (3, cout)):
        _conv_spec(out, '%sconv%d' % (prefix, j), ci, cout, 3)
        _bn_spec(out, '%sbn%d' % (prefix, j), cout, norm)
    return out


def convu_specs(planes, first, prefix='', out=None, norm='bn'):
    """networks.unet.ConvU (unet.py:75-94)."""
    out = [] if out is None else out
    if not first:
        _conv_spec(out, prefix + 'conv1', 2 * planes, planes, 3)
        _bn_spec(out, prefix + 'bn1', planes, norm)
    _conv_spec(out, prefix + 'conv2', planes, planes // 2, 1)
    _bn_spec(out, prefix + 'bn2', planes // 2, norm)
    _conv_spec(out, prefix + 'conv3', planes, planes, 3)
    _bn_spec(out, prefix + 'bn3', planes, norm)
    return out


def convu_rec_specs(planes, num_domains, prefix='', out=None):
    """networks.unet.ConvU_Rec (unet.py:120-137); DSBN when num_domains is given."""
    out = [] if out is None else out
    h = planes // 2
    _conv_spec(out, prefix + 'conv1', planes, h, 3)
    _norm_spec(out, prefix + 'bn1', h, num_domains)
    _conv_spec(out, prefix + 'conv2', h, h, 1)
    _norm_spec(out, prefix + 'bn2', h, num_domains)
    _conv_spec(out, prefix + 'conv3', h, h, 3)
    _norm_spec(out, prefix + 'bn3', h, num_domains)
    return out


def encoder_specs(c=3, n=16, norm='bn'):
    """state_dict layout of networks.unet.Encoder (unet.py:248-255): 105 entries at n=16 (norm='bn')."""
    out = []
    ch = [c, n, 2 * n, 4 * n, 8 * n, 16 * n]
    for l in range(1, 6):
        convd_specs(ch[l - 1], ch[l], 'convd%d.' % l, out, norm)
    return out


def decoder_specs(n=16, num_classes=2, norm='bn'):
    """networks.unet.Decoder (unet.py:273-281)."""
    out = []
    for l, planes, first in ((4, 16 * n, True), (3, 8 * n, False), (2, 4 * n, False), (1, 2 * n, False)):
        convu_specs(planes, first, 'convu%d.' % l, out, norm)
    _conv_spec(out, 'out1', 2 * n, num_classes, 3)
    return out


def rec_decoder_specs(n=16, num_classes=3, num_domains=None):
    """networks.unet.Rec_Decoder (unet.py:299-307); norm='dsbn' when num_domains is given."""
    out = []
    for l, planes in ((4, 16 * n), (3, 8 * n), (2, 4 * n), (1, 2 * n)):
        convu_rec_specs(planes, num_domains, 'convu%d.' % l, out)
    _conv_spec(out, 'out1', n, num_classes, 3)
    return out


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


class ParamBank:
    """Flat fp32 arenas (parameters, gradients, Adam moments) for a list of modules, in the reference's
    ``parameters()`` order, plus the BN buffers; tensors handed out are views, so an optimizer step
    on the arena is an optimizer step on every nn.Parameter."""

    def __init__(self, modules, device):
        # modules: list of (module_name, specs)
        self.device = device
        self.index = {}             # (module, key) -> (offset, shape)
        self.module_range = {}
        off = 0
        for mname, specs in modules:
            start = off
            for key, shape, kind, _ in specs:
                if kind == 'param':
                    self.index[(mname, key)] = (off, shape)
                    off += _numel(shape)
            self.module_range[mname] = (start, off)
        self.n = off
        self.params = torch.zeros(off, dtype=torch.float32, device=device)
        self.grads = torch.zeros(off, dtype=torch.float32, device=device)
        self.exp_avg = None
        self.exp_avg_sq = None
        self.buffers = {}
        for mname, specs in modules:
            for key, shape, kind, dt in specs:
                if kind == 'buffer':
                    if key.endswith('running_var'):
                        t = torch.ones(shape, dtype=dt, device=device)
                    else:
                        t = torch.zeros(shape, dtype=dt, device=device)
                    self.buffers[(mname, key)] = t

    def view(self, arena, mname, key):
        off, shape = self.index[(mname, key)]
        return arena[off:off + _numel(shape)].view(shape)

    def p(self, mname, key):
        return self.view(self.params, mname, key)

    def g(self, mname, key):
        return self.view(self.grads, mname, key)

    def b(self, mname, key):
        return self.buffers[(mname, key)]

    def ensure_adam(self):
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.params)
            self.exp_avg_sq = torch.zeros_like(self.params)


# ------------------------------------------------------------------------------------------------ graph
class Norm:
    """One normalisation site: per-group parameter views (DSBN: one BN per group; shared BN: the same
    views in every group)."""

    def __init__(self, bank, mname, name, groups_to_domain=None, kind='bn'):
        self.bank, self.mname, self.name = bank, mname, name
        self.groups_to_domain = groups_to_domain      # None: plain BN shared by all groups
        # 'bn': BatchNorm statistics per group of images; 'gn' / 'in' (unet.py:20-23): the plan has ONE GROUP PER IMAGE and the
        # statistics are pooled over (C, H, W) resp. (H, W) of the image -- rd_gn_finalize_* resp. rd_bn_finalize_* with constant
        # gamma = 1 / beta = 0; neither has running statistics, 'in' has no parameters at all
        self.kind = kind

    def key(self, g, leaf):
        if self.groups_to_domain is None:
            return '%s.%s' % (self.name, leaf)
        return '%s.bns.%d.%s' % (self.name, self.groups_to_domain[g], leaf)

    def param(self, g, leaf):
        return self.bank.p(self.mname, self.key(g, leaf))

    def grad(self, g, leaf):
        return self.bank.g(self.mname, self.key(g, leaf))

    def buf(self, g, leaf):
        return self.bank.b(self.mname, self.key(g, leaf))

    # device pointers for the finalize descriptors (None where the normalisation has no such tensor)
    def gamma_ptr(self, plan, g, C):
        return plan.const_f32(1.0, C).data_ptr() if self.kind == 'in' else self.param(g, 'weight').data_ptr()

    def beta_ptr(self, plan, g, C):
        return plan.const_f32(0.0, C).data_ptr() if self.kind == 'in' else self.param(g, 'bias').data_ptr()

    def buf_ptr(self, g, leaf):
        return self.buf(g, leaf).data_ptr() if self.kind == 'bn' else None

    def grad_ptr(self, g, leaf):
        return None if self.kind == 'in' else self.grad(g, leaf).data_ptr()


class Act:
    """A stored raw tensor plus the pending (virtual) BatchNorm + activation applied by its readers."""

    def __init__(self, plan, N, H, W, Cc, norm=None, act=False, up=False, name='', cstride=None, grad_cstride=None):
        self.plan, self.N, self.H, self.W, self.C = plan, N, H, W, Cc
        self.norm, self.act, self.up, self.name = norm, act, up, name
        # narrow tensors (3-channel image, 2/3-class dlogits) may be stored with a zero-padded channel tail so
        # that their readers fetch whole 16-byte channel vectors; producers never write the pad
        self.Cs, self.gCs = cstride or Cc, grad_cstride or Cc
        self.y_buf = plan.alloc_act((N, 2 * H, 2 * W, Cc)) if (up and plan.materialize_up) else None
        # wide layers: relu(bn(z)) and the BN-backward gradient are stored once (rd_bn_apply) for their many readers
        mat = (norm is not None and act and plan.materialize_min_c is not None and Cc >= plan.materialize_min_c and
               (not up or self.y_buf is not None))
        self.a_buf = plan.alloc_act((N, 2 * H, 2 * W, Cc) if up else (N, H, W, Cc)) if mat else None
        self.dz_buf = None
        self.buf = plan.alloc_act((N, H, W, self.Cs))
        self.g = None                 # gradient w.r.t. the BN output (hi-res when up)
        self.g_written = False
        G = plan.G
        if norm is not None:
            self.stats = plan.alloc_stat(G * L.STAT_SLOTS * Cc * 2)
            self.bstats = plan.alloc_stat(G * L.STAT_SLOTS * Cc * 2)
            self.scale, self.shift = plan.alloc_f32(G * Cc), plan.alloc_f32(G * Cc)
            self.mean, self.invstd = plan.alloc_f32(G * Cc), plan.alloc_f32(G * Cc)
            self.P, self.Q, self.R = plan.alloc_f32(G * Cc), plan.alloc_f32(G * Cc), plan.alloc_f32(G * Cc)
        else:
            self.stats = self.bstats = self.scale = self.shift = None

    def grad_buf(self):
        if self.g is None:
            Hh, Ww = (2 * self.H, 2 * self.W) if self.up else (self.H, self.W)
            self.g = self.plan.alloc_act((self.N, Hh, Ww, self.gCs))
        return self.g


class ConvNode:
    def __init__(self, mname, name, inputs, out, taps, Cin, Cout, has_bias_grad):
        self.mname, self.name, self.inputs, self.out = mname, name, inputs, out
        self.taps, self.Cin, self.Cout = taps, Cin, Cout
        self.has_bias_grad = has_bias_grad          # out1 convs: live bias (no BN behind it)


class PoolNode:
    """nn.MaxPool2d(2) in front of a ConvD (unet.py:56), materialised: `out` = maxpool2(act(bn(src))) stored once."""
    def __init__(self, mname, name, src, out):
        self.mname, self.name, self.src, self.out = mname, name, src, out


class Plan:
    """Buffers + launch lists for one batch geometry (N images in G groups at H x W)."""

    def __init__(self, bank, dtype, N, gstart, slope=0.0, training=True):
        self.bank, self.dtype = bank, dtype
        self.dt = L.RD_BF16 if dtype == torch.bfloat16 else L.RD_F32
        self.device = bank.device
        self.N, self.gstart, self.G = N, list(gstart), len(gstart) - 1
        self.slope, self.training = slope, training
        self.pad_narrow = False       # TrainStep: pad the dlogits of the output convs to one 16-byte slot
        # True: rd_up_stats also stores y = up2(t), and the 3x3 conv behind it (forward, dgrad epilogue, wgrad)
        # reads y as a plain BN+ReLU source instead of interpolating t four-taps-per-pixel in every loader:
        # +1 write / +3 reads of y against ~40% less time in those three kernels ('upsample')
        self.materialize_up = False
        # True: the 2x2 max-pool in front of ConvD levels 2-5 is stored once (rd_pool_fwd, 1/4 of the pixels) and its
        # gradient scattered by rd_pool_bwd, so the conv behind it runs the plain-source kernels in forward, dgrad and
        # wgrad (level 2 at 400x400, 16 images: 576 -> ~230 us for the three launches; profiles/README.md round 2).
        # False: RD_SRC_POOL / RD_DST_POOL fused into the conv's tile loader / gradient epilogue.
        self.materialize_pool = T.options()['pool_mat']
        self.side_cus = 0               # >0: compute-unit budget of the weight-gradient launches (they run on a side stream)
        self.conv_cus = 0               # >0: compute-unit budget of this plan's persistent conv launches (a side-lane plan)
        self.dgrad_cus = 0              # >0: ... of its >= 64-channel gradient launches only (tuning.py dgrad_cus)
        self.materialize_min_c = None   # channels from which BN+ReLU outputs are stored once (rd_bn_apply)
        self.materialize_dz_min_c = None  # ... and from which the BN-backward gradients dz are
        self.materialize_dz_wide = bool(T.options()['mat_dz_wide'])
        self.fused_bwd = bool(T.options()['fused_bwd'])   # small-channel 3x3 convs: dgrad + weight gradient in one launch
        # True: a conv launch that runs on the warp-specialised 64-wide kernel also STORES what its loader waves stage (rd_src_t.out:
        # act(bn(z)) in the forward launch, dz in the gradient launch), and the layer's weight gradient reads those stored operands
        # (RD_SRC_RAW: its loader copies instead of repeating the transforms, and the loader is that kernel's pole)
        self.store_operands = int(T.options()['store_wgrad_operands'])     # bit 0: the forward operand, bit 1: dz
        self.store_min_c = int(T.options()['store_wgrad_min_c'])           # ... of tensors with at least this many channels only
        self.split_wide_dgrad = bool(T.options()['split_wide_dgrad'])   # _dgrad_halves below
        # True (the fused training step): no BatchNorm finalize launches between the convs -- the launch that first reads a layer's
        # coefficients derives them from the statistic slots in its prologue (rd_src_t.fin, csrc/bn_fin.h), and every launch spreads
        # its sums over RD_STAT_SLOTS_FOLD copies only.  False: explicit rd_bn_finalize_* launches (modules, eval plans, gn / in)
        self.fold_finalize = False
        self.fold_wgrad_behind = bool(T.options()['fold_wgrad_behind'])
        self.fold_fwd_kinds = int(T.options()['fold_fwd_kinds'])
        self._unit = {}
        self.nodes = []
        self.keep = []
        self._stat_chunks = []
        self.fwd, self.bwd = [], []
        self.ws_bytes = 0
        if self.G > L.MAXG:
            raise ValueError('%d statistics groups (gn / in: one per image), the kernels take at most %d' % (self.G, L.MAXG))
        self.gs_arr = L.gstart_array(self.gstart)
        self._consts = {}

    # ---- allocation
    def const_f32(self, value, n):
        key = (float(value), int(n))
        if key not in self._consts:
            self._consts[key] = torch.full((n,), float(value), dtype=torch.float32, device=self.device)
        return self._consts[key]

    def alloc_act(self, shape):
        t = torch.zeros(shape, dtype=self.dtype, device=self.device)
        self.keep.append(t)
        return t

    def alloc_f32(self, n):
        t = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.keep.append(t)
        return t

    def alloc_stat(self, n):
        self._stat_chunks.append(n)
        return ('stat', len(self._stat_chunks) - 1)

    def finalize_stats(self):
        total = sum(self._stat_chunks)
        # fp64: the slot atomics and everything the finalize kernels do across workgroups (include/ramdsir.h, RD_STAT_SLOTS)
        self.stat_arena = torch.zeros(max(total, 1), dtype=torch.float64, device=self.device)
        offs, o = [], 0
        for n in self._stat_chunks:
            offs.append(o)
            o += n
        self._stat_off = offs

    def stat_ptr(self, handle):
        return self.stat_arena.data_ptr() + 8 * self._stat_off[handle[1]]

    def stat_view(self, handle, G, Cc):
        o = self._stat_off[handle[1]]
        return self.stat_arena[o:o + G * L.STAT_SLOTS * Cc * 2].view(G, L.STAT_SLOTS, Cc, 2).sum(1)

    # ---- graph construction
    def conv(self, mname, name, inputs, Cout, taps, norm=None, act=False, up_out=False, H=None, W=None, N=None):
        """inputs: list of (Act, mode, n_off, g_fixed).  Output dims: H x W of the conv itself."""
        if self.materialize_pool:
            pooled = []
            for (a, mode, n_off, g_fixed) in inputs:
                if mode == L.SRC_POOL and a.C % self.slot_channels() == 0:
                    xp = Act(self, a.N, a.H // 2, a.W // 2, a.C, name='%s.%s.pool' % (mname, name))
                    xp.needs_grad = True
                    self.nodes.append(PoolNode(mname, name + '.pool', a, xp))
                    pooled.append((xp, L.SRC_RAW, n_off, g_fixed))
                else:
                    pooled.append((a, mode, n_off, g_fixed))
            inputs = pooled
        Cin = sum(a.C for a, _, _, _ in inputs)
        gcs = self.slot_channels() if (self.pad_narrow and norm is None and Cout < self.slot_channels()) else None
        out = Act(self, N if N is not None else self.N, H, W, Cout, norm=norm, act=act, up=up_out, name='%s.%s' % (mname, name),
                  grad_cstride=gcs)
        node = ConvNode(mname, name, inputs, out, taps, Cin, Cout, norm is None)
        self.nodes.append(node)
        return out

    def unit_coef(self, Cc):
        """(ones, zeros) [G][C] coefficient rows: identity BN for readers of an already materialised tensor."""
        if Cc not in self._unit:
            self._unit[Cc] = (self.alloc_f32(self.G * Cc).fill_(1.0), self.alloc_f32(self.G * Cc))
        return self._unit[Cc]

    def slot_channels(self):
        return 8 if self.dtype == torch.bfloat16 else 4

    # ---- folded BatchNorm finalize (rd_src_t.fin)
    def stat_slots(self):
        """rd_conv_t.stat_slots / the stat_slots arguments of this plan's launches (0 = all RD_STAT_SLOTS copies)."""
        if os.environ.get('RAMDSIR_DEBUG_LIB') == '1' and 'RD_FOLD_SLOTS' in os.environ:      # experiments: how many copies a folded producer spreads its sums over
            n = int(os.environ['RD_FOLD_SLOTS'])
            if n < 0 or n > L.STAT_SLOTS or n % 8:              # the folded prologues read the copies in blocks of eight (csrc/bn_fin.h slot_sums)
                raise ValueError('RD_FOLD_SLOTS must be 0 or a multiple of 8 up to %d' % L.STAT_SLOTS)
            return n
        return L.STAT_SLOTS_FOLD if self.fold_finalize else 0

    def _folds(self, o, direction='fwd'):
        """The finalize launches of Act `o` are folded into their consumers: BatchNorm in training mode, coefficients read through
        scale / shift resp. P / Q / R by launches that take a `fin` (not the materialising rd_bn_apply passes).  fold_finalize: 1 both
        directions, 3 the forward finalizes only, 4 the backward ones only (2: timing experiment, none at all)."""
        if o.plan.fold_finalize == (4 if direction == 'fwd' else 3):      # (5 / 6: Plan.build un-folds part of the backward ones)
            return False
        return bool(o.plan.fold_finalize and o.plan.training and o.norm is not None and o.norm.kind == 'bn' and o.a_buf is None
                    and o.plan.materialize_dz_min_c is None and not o.plan.split_wide_dgrad and o.plan.G <= L.FIN_MAX_G)

    def _fin_ptr(self, desc):
        """rd_src_t.fin of a consumer: the address of the (host) finalize descriptor, which the entry point copies into the kernel
        arguments at launch time.  The descriptor is kept alive by Plan.keep."""
        return C.addressof(desc)

    def _take_fwd_fin(self, a, kind=1):
        """(fin, flags) for a FORWARD launch of this plan that reads Act a's scale / shift.  The first consumer inside the owning plan
        is the owner and later ones (same stream, behind it) need nothing; a consumer in ANOTHER plan runs on another lane, possibly
        beside the owner, so it derives the coefficients itself (restoration decoder on the encoder bottleneck)."""
        if getattr(a, 'fin_fwd', None) is None or self.fold_finalize == 2:      # 2: timing experiment -- no finalize at all (wrong results)
            return None, 0
        if not (self.fold_fwd_kinds & kind) and a.plan is self and not a.fin_fwd_taken:
            # this kind of consumer (1 small-channel conv, 2 wide conv, 4 max-pool: tuning.py fold_fwd_kinds) keeps the explicit launch, in front of it
            a.fin_fwd_taken = True
            self.fwd.append(a.fin_fwd_op)
            return None, 0
        if a.plan is not self:
            if a.plan.stat_slots() != self.stat_slots():
                raise ValueError('plans that share a BatchNorm site must agree on fold_finalize')
            return a.fin_fwd, 0
        if a.fin_fwd_taken:
            return None, 0
        a.fin_fwd_taken = True
        return a.fin_fwd, L.FIN_OWNER

    # ---- descriptor helpers
    def _src(self, a, mode, n_off, g_fixed, fold=0):
        """fold != 0: this is a forward launch's source -- attach the folded BatchNorm finalize of `a` if it is still pending (the value
        is the consumer's kind for tuning.py fold_fwd_kinds: 1 small-channel conv, 2 wide conv)."""
        s = L.RdSrc()
        slope = self.slope
        if a.a_buf is not None and mode in (L.SRC_AFFACT, L.SRC_UP, L.SRC_POOL):
            # stored relu(bn(.)): read as is; the pooled read keeps its 2x2 max with identity coefficients
            s.ptr = a.a_buf.data_ptr()
            if mode == L.SRC_POOL:
                one, zero = self.unit_coef(a.C)
                s.scale, s.shift, slope = one.data_ptr(), zero.data_ptr(), 1.0
                g_fixed = 0
            else:
                mode = L.SRC_RAW
        elif a.norm is None and mode == L.SRC_POOL:
            # 2x2 max-pool of a tensor that has no pending BatchNorm (ConvD called on its own): identity coefficients
            one, zero = self.unit_coef(a.C)
            s.ptr, s.scale, s.shift, slope, g_fixed = a.buf.data_ptr(), one.data_ptr(), zero.data_ptr(), 1.0, 0
        else:
            if mode == L.SRC_UP and a.y_buf is not None:
                s.ptr, mode = a.y_buf.data_ptr(), L.SRC_AFFACT
            else:
                s.ptr = a.buf.data_ptr()
            if mode != L.SRC_RAW:
                s.scale, s.shift = a.scale.data_ptr(), a.shift.data_ptr()
                if fold:
                    fin, flags = self._take_fwd_fin(a, fold)
                    if fin is not None:
                        s.fin, s.fin_flags = fin, flags
        s.mode, s.C, s.slope, s.n_off, s.g_fixed = mode, a.Cs, slope, n_off, g_fixed
        return s

    def _dz_src(self, node, owner=None):
        """Source descriptor of the gradient w.r.t. this conv's output.  owner (True / False; None: no fin): the launch carries the
        folded BatchNorm-backward finalize of the site, as its owner (the gradient launch) or not (the weight gradient beside it)."""
        o = node.out
        s = L.RdSrc()
        s.C, s.slope, s.n_off, s.g_fixed = (o.gCs if o.norm is None else o.C), 0.0, 0, -1
        if o.dz_buf is not None:
            s.ptr, s.mode = o.dz_buf.data_ptr(), L.SRC_RAW
        elif o.norm is None or o.up:
            s.ptr, s.mode = (o.dt_buf if o.up else o.grad_buf()).data_ptr(), L.SRC_RAW
        else:
            s.ptr, s.ptr2, s.mode = o.grad_buf().data_ptr(), o.buf.data_ptr(), L.SRC_BNBWD
            s.scale, s.shift, s.q = o.P.data_ptr(), o.R.data_ptr(), o.Q.data_ptr()
            if owner is not None and getattr(o, 'fin_bwd', None) is not None and self.fold_finalize != 2:
                s.fin, s.fin_flags = o.fin_bwd, (L.FIN_OWNER if owner else 0)
        return s

    def build(self, wpack):
        """wpack: WeightPack giving packed-weight pointers per (module, conv name)."""
        lib = L.lib()
        self.finalize_stats()
        dt = self.dt
        nsl = self.stat_slots()
        ws_need = 0
        self.fwd_split = {}
        for node in self.nodes:
            o = node.out
            H, W, N = o.H, o.W, o.N
            if node.mname not in self.fwd_split:
                self.fwd_split[node.mname] = len(self.fwd)
            if isinstance(node, PoolNode):
                a = node.src
                has_bn = a.norm is not None
                fin, fin_flags = self._take_fwd_fin(a, 4) if has_bn else (None, 0)
                self.fwd.append((lib.rd_pool_fwd, (a.buf.data_ptr(), a.scale.data_ptr() if has_bn else None, a.shift.data_ptr() if has_bn else None,
                                                   self.slope if (has_bn and a.act) else 1.0, o.buf.data_ptr(), N, H, W, o.C, self.G, self.gs_arr, dt,
                                                   fin, fin_flags),
                                 dict(kernel='pool', what='fwd', layer='%s.%s' % (node.mname, node.name))))
                continue
            # ---------------- forward conv
            p = L.RdConv()
            p.cu_limit = int(self.conv_cus)
            p.stat_slots = nsl
            small = node.Cin <= 32 and node.Cout <= 32
            for i, (a, mode, n_off, g_fixed) in enumerate(node.inputs):
                p.src[i] = self._src(a, mode, n_off, g_fixed, fold=1 if small else 2)
            p.nsrc, p.taps = len(node.inputs), node.taps
            p.w = wpack.ptr(node.mname, node.name, False)
            p.bias = self.bank.p(node.mname, node.name + '.bias').data_ptr()
            p.CinPad, p.CoutPad = wpack.pads(node.Cout, node.Cin)
            p.N, p.H, p.W, p.Cin, p.Cout = N, H, W, node.Cin, node.Cout
            p.G, p.gstart = self.G, self.gs_arr
            p.emode, p.out = 0, o.buf.data_ptr()
            p.stats = o.plan.stat_ptr(o.stats) if (o.norm is not None and not o.up) else None
            node.a_store = [None] * len(node.inputs)
            if ((self.store_operands & 1) and self.training and self.dtype == torch.bfloat16 and node.taps == 9 and node.Cin >= self.store_min_c
                    and lib.rd_conv_honours_src_out(C.byref(p), dt)):
                for i, (a, mode, n_off, g_fixed) in enumerate(node.inputs):
                    if p.src[i].mode in (L.SRC_AFF, L.SRC_AFFACT):
                        src_t = a.y_buf if (mode == L.SRC_UP and a.y_buf is not None) else a.buf
                        node.a_store[i] = self.alloc_act(tuple(src_t.shape))
                        p.src[i].out = node.a_store[i].data_ptr()
            self.keep.append(p)
            self.fwd.append((lib.rd_conv, (C.byref(p), dt), self._conv_meta(node, N, H, W, node.Cin, node.Cout, 'fwd')))
            if o.norm is not None:
                if o.up:
                    self.fwd.append((lib.rd_up_stats, (o.buf.data_ptr(), o.plan.stat_ptr(o.stats), o.y_buf.data_ptr() if o.y_buf is not None else None,
                                                        N, H, W, o.C, self.G, self.gs_arr, dt, nsl)))
                b = L.RdBnFwd()
                b.stats = o.plan.stat_ptr(o.stats)
                # the conv epilogues sum the result WITHOUT its bias; rd_up_stats sums y = up2(t) itself
                b.conv_bias = None if o.up else self.bank.p(node.mname, node.name + '.bias').data_ptr()
                b.scale, b.shift, b.mean, b.invstd = o.scale.data_ptr(), o.shift.data_ptr(), o.mean.data_ptr(), o.invstd.data_ptr()
                hw = (4 if o.up else 1) * H * W
                kind = o.norm.kind
                if kind != 'bn' and any(self.gstart[g + 1] - self.gstart[g] != 1 for g in range(self.G)):
                    raise ValueError('norm=%r needs a launch plan with one statistics group per image (gstart = 0..N)' % kind)
                for g in range(self.G):
                    b.gamma[g] = o.norm.gamma_ptr(self, g, o.C)
                    b.beta[g] = o.norm.beta_ptr(self, g, o.C)
                    b.running_mean[g] = o.norm.buf_ptr(g, 'running_mean')
                    b.running_var[g] = o.norm.buf_ptr(g, 'running_var')
                    b.num_batches_tracked[g] = o.norm.buf_ptr(g, 'num_batches_tracked')
                    b.count[g] = float((self.gstart[g + 1] - self.gstart[g]) * hw)
                # gn / in always normalise with the statistics of the input itself (nn.GroupNorm / nn.InstanceNorm2d in eval mode too)
                b.C, b.G, b.eps, b.momentum, b.training = o.C, self.G, EPS, MOMENTUM, 1 if (self.training or kind != 'bn') else 0
                b.nslots = nsl
                self.keep.append(b)
                o.bn_desc = b
                if self._folds(o):
                    o.fin_fwd, o.fin_fwd_taken = self._fin_ptr(b), False          # taken by the first launch that reads scale / shift
                    o.fin_fwd_op = (lib.rd_bn_finalize_fwd, (C.byref(b),))
                else:
                    self.fwd.append((lib.rd_gn_finalize_fwd if kind == 'gn' else lib.rd_bn_finalize_fwd, (C.byref(b),)))
                if o.a_buf is not None:
                    src = o.y_buf if o.up else o.buf
                    Hh, Ww = (2 * H, 2 * W) if o.up else (H, W)
                    self.fwd.append((lib.rd_bn_apply, (src.data_ptr(), None, o.a_buf.data_ptr(), o.scale.data_ptr(), None, o.shift.data_ptr(),
                                                       self.slope, N, Hh, Ww, o.C, self.G, self.gs_arr, dt)))
        # a folded forward finalize exists only inside the launch that TAKES it (_take_fwd_fin, as owner).  A BatchNorm site whose readers are
        # all in other plans (flags 0), materialising passes or consumer kinds without a `fin` would otherwise never write its saved mean /
        # invstd, running statistics and num_batches_tracked -- and the backward pass would read stale values: the explicit launch, at the end
        for node in self.nodes:
            o = node.out
            if getattr(o, 'fin_fwd', None) is not None and o.plan is self and not o.fin_fwd_taken and self.fold_finalize != 2:
                o.fin_fwd_taken = True
                self.fwd.append(o.fin_fwd_op)
        if not self.training:
            return
        # ---------------- backward, reverse order; bwd_split[m] = index where module m's backward starts
        self.bwd_split, self.bwd_node_start = {}, {}
        for node in reversed(self.nodes):
            if node.mname not in self.bwd_split:
                self.bwd_split[node.mname] = len(self.bwd)
            self.bwd_node_start[(node.mname, node.name)] = len(self.bwd)
            o = node.out
            H, W, N = o.H, o.W, o.N
            if isinstance(node, PoolNode):
                a = node.src
                has_bn = a.norm is not None
                if has_bn or getattr(a, 'needs_grad', False):
                    self.bwd.append((lib.rd_pool_bwd, (o.grad_buf().data_ptr(), a.buf.data_ptr(), a.scale.data_ptr() if has_bn else None,
                                                       a.shift.data_ptr() if has_bn else None, self.slope if (has_bn and a.act) else 1.0,
                                                       1 if (has_bn and a.act) else 0, a.grad_buf().data_ptr(), 1 if a.g_written else 0,
                                                       a.plan.stat_ptr(a.bstats) if has_bn else None, N, H, W, o.C, self.G, self.gs_arr, dt, nsl),
                                     dict(kernel='pool', what='bwd', layer='%s.%s' % (node.mname, node.name))))
                    a.g_written = True
                continue
            if o.norm is not None:
                q = L.RdBnBwd()
                q.bstats = o.plan.stat_ptr(o.bstats)
                q.mean, q.invstd = o.mean.data_ptr(), o.invstd.data_ptr()
                q.P, q.Q, q.R = o.P.data_ptr(), o.Q.data_ptr(), o.R.data_ptr()
                hw = (4 if o.up else 1) * H * W
                for g in range(self.G):
                    q.gamma[g] = o.norm.gamma_ptr(self, g, o.C)
                    q.dgamma[g] = o.norm.grad_ptr(g, 'weight')
                    q.dbeta[g] = o.norm.grad_ptr(g, 'bias')
                    q.count[g] = float((self.gstart[g + 1] - self.gstart[g]) * hw)
                q.C, q.G = o.C, self.G
                if o.norm.kind == 'gn':                        # GroupNorm: the conv bias in front of it has a gradient (ramdsir.h)
                    q.fstats = o.plan.stat_ptr(o.stats)
                    q.conv_bias = None if o.up else self.bank.p(node.mname, node.name + '.bias').data_ptr()
                    q.dbias = self.bank.g(node.mname, node.name + '.bias').data_ptr()
                q.nslots = nsl
                self.keep.append(q)
                fin_op, fin_pos = (lib.rd_bn_finalize_bwd, (C.byref(q),)), len(self.bwd)
                if self._folds(o, 'bwd') and not (o.up and o.plan.fold_finalize == 6):
                    o.fin_bwd = self._fin_ptr(q)            # carried by rd_up_bwd, or by the gradient launch (owner) and the weight gradient
                else:
                    self.bwd.append((lib.rd_gn_finalize_bwd if o.norm.kind == 'gn' else lib.rd_bn_finalize_bwd, (C.byref(q),)))
                # dz stored once: every >= 64-channel layer, and (round 4) a 32-channel layer whose gradient launch is 64-wide (more
                # than 32 input channels: dec.convu1.conv1, rec.convu2.conv1) -- a BatchNorm-backward source keeps that launch on the
                # two-operand conv_pf_kernel (110 us, the slowest gradient launch of the step); with a stored dz it runs on the
                # persistent conv_ws_kernel like the other 64-wide gradients
                wide_dgrad = self.materialize_dz_wide and isinstance(node, ConvNode) and node.taps == 9 and node.Cin > 32 and o.C >= 32
                if (not o.up and self.materialize_dz_min_c is not None and (o.C >= self.materialize_dz_min_c or wide_dgrad)):
                    o.dz_buf = self.alloc_act((N, H, W, o.C))
                    self.bwd.append((lib.rd_bn_apply, (o.grad_buf().data_ptr(), o.buf.data_ptr(), o.dz_buf.data_ptr(), o.P.data_ptr(),
                                                       o.Q.data_ptr(), o.R.data_ptr(), 1.0, N, H, W, o.C, self.G, self.gs_arr, dt)))
                if o.up:
                    o.dt_buf = self.alloc_act((N, H, W, o.C))
                    fb = getattr(o, 'fin_bwd', None) if self.fold_finalize != 2 else None
                    self.bwd.append((lib.rd_up_bwd, (o.grad_buf().data_ptr(), o.buf.data_ptr(), o.dt_buf.data_ptr(), o.P.data_ptr(),
                                                     o.Q.data_ptr(), o.R.data_ptr(), N, H, W, o.C, self.G, self.gs_arr, dt,
                                                     fb, L.FIN_OWNER if fb is not None else 0)))
            # wgrad descriptor (launched below: fused with the dgrad where the pair qualifies)
            wg = L.RdWgrad()
            for i, (a, mode, n_off, g_fixed) in enumerate(node.inputs):
                wg.a[i] = self._src(a, mode, n_off, g_fixed)
                if getattr(node, 'a_store', None) and node.a_store[i] is not None:      # stored by the forward launch (rd_src_t.out)
                    wg.a[i].ptr, wg.a[i].mode, wg.a[i].scale, wg.a[i].shift = node.a_store[i].data_ptr(), L.SRC_RAW, None, None
            wg.na, wg.taps = len(node.inputs), node.taps
            wg.dz = self._dz_src(node, owner=False)          # (promoted to owner below when the conv has no gradient launch)
            wg.N, wg.H, wg.W, wg.Cin, wg.Cout = N, H, W, node.Cin, node.Cout
            wg.G, wg.gstart = self.G, self.gs_arr
            wg.dW = self.bank.g(node.mname, node.name + '.weight').data_ptr()
            wg.beta = 0.0
            wg.cu_limit = int(self.side_cus)
            self.keep.append(wg)
            node.wg = wg
            # algorithmic work of the weight gradient (SURVEY.md 8d convention): the conv's logical input read once + the
            # gradient w.r.t. its output read once (a BN-backward source is two tensors: g and the raw z); dW is negligible
            esz = 2 if self.dtype == torch.bfloat16 else 4
            dz_ops = 2 if wg.dz.mode == L.SRC_BNBWD else 1
            dz_c = o.gCs if o.norm is None else o.C
            wg_bytes, wg_flops = N * H * W * (node.Cin + dz_ops * dz_c) * esz, 2 * N * H * W * node.Cin * node.Cout * node.taps
            # dgrad descriptor (none when no input needs a gradient, i.e. the first conv on the image)
            dsts = []
            for (a, mode, n_off, g_fixed) in node.inputs:
                d = L.RdDst()
                if mode == L.SRC_RAW and a.norm is None and not getattr(a, 'needs_grad', False):
                    d.kind = L.DST_NONE
                else:
                    d.kind = {L.SRC_RAW: L.DST_PLAIN, L.SRC_AFF: L.DST_PLAIN, L.SRC_AFFACT: L.DST_PLAIN,
                              L.SRC_POOL: L.DST_POOL, L.SRC_UP: L.DST_UPY}[mode]
                    d.g = a.grad_buf().data_ptr()
                    if d.kind == L.DST_UPY and a.y_buf is not None:
                        d.kind = L.DST_PLAIN
                    if a.norm is not None:
                        d.z = (a.y_buf if (mode == L.SRC_UP and a.y_buf is not None) else a.buf).data_ptr()
                        d.scale, d.shift = a.scale.data_ptr(), a.shift.data_ptr()
                        d.bstats = a.plan.stat_ptr(a.bstats)
                    d.act = 1 if (a.act and a.norm is not None) else 0
                    d.accumulate = 1 if a.g_written else 0
                    a.g_written = True
                    d.Cd, d.slope, d.n_off, d.g_fixed = a.C, self.slope, n_off, g_fixed
                    if a.norm is None and d.kind == L.DST_POOL:
                        # the arg-max of the scatter needs the pooled values: the raw tensor under identity coefficients
                        one, zero = self.unit_coef(a.C)
                        d.z, d.scale, d.shift, d.slope, d.g_fixed = a.buf.data_ptr(), one.data_ptr(), zero.data_ptr(), 1.0, 0
                dsts.append(d)
            p = None
            if not all(d.kind == L.DST_NONE for d in dsts):
                p = L.RdConv()
                p.cu_limit = int(self.conv_cus)
                p.stat_slots = nsl
                if not p.cu_limit and self.dgrad_cus and (node.Cin > 32 or node.Cout > 32):
                    p.cu_limit = int(self.dgrad_cus)
                p.src[0] = self._dz_src(node, owner=True)
                p.nsrc, p.taps = 1, node.taps
                p.w = wpack.ptr(node.mname, node.name, True)
                p.bias = None
                p.CinPad, p.CoutPad = wpack.pads(node.Cin, node.Cout)       # roles swapped: K = Cout, N = Cin
                p.N, p.H, p.W, p.Cin, p.Cout = N, H, W, node.Cout, node.Cin
                p.G, p.gstart = self.G, self.gs_arr
                p.emode = 1
                p.dst[0] = dsts[0]
                if len(dsts) == 2:
                    p.dst[1] = dsts[1]
                    p.c_split = node.inputs[0][0].C
                else:
                    p.dst[1].kind = L.DST_NONE
                    p.c_split = node.Cin
                self.keep.append(p)
            if p is None and wg.dz.fin:
                wg.dz.fin_flags = L.FIN_OWNER                # the first conv on the image: its weight gradient is the only reader of P / Q / R
            dmeta = self._conv_meta(node, N, H, W, node.Cout, node.Cin, 'dgrad') if p is not None else None
            node.fused = bool(p is not None and self.fused_bwd and lib.rd_conv_bwd_fused_ok(C.byref(p), C.byref(wg), dt))
            if self.fold_finalize in (5, 6) and not node.fused and not o.up and getattr(o, 'fin_bwd', None) is not None:
                # fold_finalize 5 / 6: the backward finalize is folded only into launches that do gradient AND weight gradient at once
                # (and, 5, into rd_up_bwd); a layer with two stand-alone launches keeps the explicit one in front of both
                o.fin_bwd = None
                wg.dz.fin, wg.dz.fin_flags = None, 0
                if p is not None:
                    p.src[0].fin, p.src[0].fin_flags = None, 0
                assert fin_pos >= self.bwd_node_start[(node.mname, node.name)]      # inside this node's range: no recorded index moves
                self.bwd.insert(fin_pos, fin_op)
            dgrad_first = False
            if node.fused:
                # small-channel 3x3 conv: dgrad + weight gradient in ONE launch on the dgrad chain (csrc/conv_fused.hip: both are
                # HBM-bound and read the same tensors); its per-workgroup dW sums are reduced on the weight-gradient lane, from a
                # buffer of this layer's own (the next fused launch of the chain must not overwrite sums still to be reduced)
                node.fused_ws = self.alloc_f32(max(lib.rd_conv_bwd_fused_workspace(C.byref(p), C.byref(wg), dt) // 4, 1))
                wg.partial = node.fused_ws.data_ptr()
                fmeta = dict(dmeta, kernel='conv_small_bwd_fused', what='dgrad+wgrad', bytes=dmeta['bytes'] + wg_bytes, flops=dmeta['flops'] + wg_flops)
                self.bwd.append((lib.rd_conv_bwd_fused, (C.byref(p), C.byref(wg), dt), fmeta))
                node.side_meta = [dict(kernel='wgrad_reduce', side=True, side_idx=0, layer='%s.%s' % (node.mname, node.name))]
                self.bwd.append((lib.rd_conv_bwd_fused_reduce, (C.byref(p), C.byref(wg), dt), node.side_meta[0]))
            else:
                # the gradient launch stores the dz its loader forms (rd_src_t.out) and the weight gradient, enqueued BEHIND it, reads that
                dgrad_first = bool(p is not None and (self.store_operands & 2) and self.dtype == torch.bfloat16 and p.src[0].mode == L.SRC_BNBWD and node.Cout >= self.store_min_c
                                   and not self.split_wide_dgrad and lib.rd_conv_honours_src_out(C.byref(p), dt))
                # a folded BatchNorm-backward finalize (fold_wgrad_order 1): the gradient launch derives P / Q / R as the owner and the weight
                # gradient, enqueued BEHIND it, reads them like any later launch -- instead of repeating the prologue on its own lane
                fold_first = bool(p is not None and not dgrad_first and p.src[0].fin and self.fold_wgrad_behind and node is not self.nodes[0])   # (the
                # plan's LAST gradient launch stays last: TrainStep moves the restoration decoder's to the main lane, behind the join)
                if fold_first:
                    wg.dz.fin, wg.dz.fin_flags = None, 0
                    self.bwd.append((lib.rd_conv, (C.byref(p), dt), dmeta))
                    dgrad_first = True
                elif dgrad_first:
                    o.dz_store = self.alloc_act((N, H, W, o.C))
                    p.src[0].out = o.dz_store.data_ptr()
                    wg.dz.ptr, wg.dz.ptr2, wg.dz.mode = o.dz_store.data_ptr(), None, L.SRC_RAW
                    wg.dz.scale = wg.dz.shift = wg.dz.q = None
                    wg.dz.fin, wg.dz.fin_flags = None, 0
                    self.bwd.append((lib.rd_conv, (C.byref(p), dt), dmeta))
                ws_need = max(ws_need, lib.rd_wgrad_workspace(C.byref(wg), dt))
                node.side_meta = [dict(kernel='wgrad', side=True, side_idx=0, layer='%s.%s' % (node.mname, node.name), bytes=wg_bytes, flops=wg_flops)]
                self.bwd.append((lib.rd_wgrad, (C.byref(wg), dt), node.side_meta[0]))
            if node.has_bias_grad:
                node.bias_ws = self.alloc_f32(8192)
                node.side_meta.append(dict(kernel='colsum', side=True, side_idx=0))
                self.bwd.append((lib.rd_colsum, (o.grad_buf().data_ptr(), self.bank.g(node.mname, node.name + '.bias').data_ptr(),
                                                 node.bias_ws.data_ptr(), N * H * W, o.C, o.gCs, 0.0, dt), node.side_meta[-1]))
            if p is not None and not node.fused and not dgrad_first:
                halves = self._dgrad_halves(p) if self.split_wide_dgrad else None
                if halves is not None:
                    for q in halves:
                        self.bwd.append((lib.rd_conv, (C.byref(q), dt), dict(dmeta, kernel='conv_small_kernel<bf16,%d,*>' % p.taps, bytes=dmeta['bytes'] // 2,
                                                                               flops=dmeta['flops'] // 2)))
                else:
                    self.bwd.append((lib.rd_conv, (C.byref(p), dt), dmeta))
        self.ws_bytes = ws_need

    def _dgrad_halves(self, p):
        """A gradient launch with ONE input chunk (<= 32 channels of dz) and 33..64 output channels into one plain tensor -- in the
        U-Net: dec.convu1.conv1 -- as TWO launches of the small-channel kernel over the 32-channel halves of the output (each reads
        dz in full): the generic 64-wide path runs it as a one-chunk K loop per workgroup with nothing to pipeline (155 us against
        ~2 x 50).  The halves point into the same packed weights (rd_conv_t.w_tap_rows).  None when the launch does not qualify."""
        if self.dtype != torch.bfloat16 or p.CinPad != 32 or p.CoutPad != 64 or p.Cout <= 32:
            return None
        d = p.dst[0]
        if p.dst[1].kind != L.DST_NONE or d.kind != L.DST_PLAIN or p.c_split < p.Cout or d.Cd % 8 or d.Cd < p.Cout:
            return None
        esz, out = 2, []
        for k in range(2):
            q = L.RdConv()
            C.memmove(C.byref(q), C.byref(p), C.sizeof(L.RdConv))
            q.Cout, q.CoutPad, q.c_split = min(32, p.Cout - 32 * k), 32, min(32, p.Cout - 32 * k)
            q.w_tap_rows = p.CoutPad
            q.w = p.w + 32 * k * p.CinPad * esz                    # rows 32k.. of every tap: [chunk][tap][CoutPad][CK]
            dd = q.dst[0]
            off = 32 * k
            dd.g = d.g + off * esz
            if d.z:
                dd.z = d.z + off * esz
            if d.scale:
                dd.scale, dd.shift = d.scale + 4 * off, d.shift + 4 * off   # fp32 [G][Cd]: the channel offset inside a row
            if d.bstats:
                dd.bstats = d.bstats + 16 * off                            # fp64 [G][slots][Cd][2]
            self.keep.append(q)
            out.append(q)
        return out

    def _conv_meta(self, node, N, H, W, Cin, Cout, what):
        """Which conv_kernel instantiation a launch uses and its ALGORITHMIC bytes: the logical input read
        once + the output written once, in the storage dtype."""
        esz = 2 if self.dtype == torch.bfloat16 else 4
        nb = 2 if ((Cout + 31) // 32 * 32) % 64 == 0 else 1
        if nb == 2 and self.dtype == torch.bfloat16:
            # csrc/conv_big.hip rd_conv_big_dispatch: grids below RD_CONV_NB1_BELOW 64-wide workgroups run 32-wide tiles
            wgs64 = ((W + 31) // 32) * ((H + 7) // 8) * N * (((Cout + 31) // 32 * 32) // 64)
            if wgs64 < T.options()['conv_nb1_below']:
                nb = 1
        tname = 'bf16' if self.dtype == torch.bfloat16 else 'f32'
        ck = 32 if self.dtype == torch.bfloat16 else 16
        if Cin <= ck and Cout <= 32:
            kname = 'conv_small_kernel<%s,%d,*>' % (tname, node.taps)       # all SRCG/EPI instantiations of the template
        else:
            kname = 'conv_kernel<%s,%d,%d>' % (tname, node.taps, nb)
        return dict(kernel=kname, what=what, layer='%s.%s' % (node.mname, node.name),
                    bytes=N * H * W * (Cin + Cout) * esz, flops=2 * N * H * W * Cin * Cout * node.taps)

    def bind_workspace(self, ws):
        """ws: float32 tensor of at least ws_bytes/4 elements, or a list of them: the weight-gradient launches then
        alternate between the workspaces, and carry side_idx = which one, so that launches on different side streams
        never share a split workspace (Plan.run_lanes sends side_idx k to lane 'side<k>')."""
        wss = list(ws) if isinstance(ws, (list, tuple)) else [ws]
        k = 0
        for node in reversed(self.nodes):                 # backward order = launch order
            if hasattr(node, 'wg'):
                if not getattr(node, 'fused', False):            # fused layers keep their own per-layer buffer
                    node.wg.partial = wss[k % len(wss)].data_ptr()
                for m in node.side_meta:
                    m['side_idx'] = k % len(wss)
                k += 1

    @staticmethod
    def run(ops, stream):
        for op in ops:
            fn, args = op[0], op[1]
            err = fn(*args, stream)
            if err:
                raise RuntimeError('ramdsir HIP launch failed: %s -> %d' % (fn.__name__, err))

    @staticmethod
    def run_lanes(ops, main, lanes, wrap=None):
        """Launch `ops` in list order over up to three HIP streams.  `lanes` maps lane names to torch streams:
          'side<k>' -- ops tagged side=True (weight gradients, bias column sums: nothing in the backward chain depends
                    on them) with side_idx k go there, each behind an event recorded at its position in the main stream
                    (two side streams: the split reduction of one layer runs beside the MFMA kernel of the next);
          'rec'  -- ops tagged lane='rec' (the restoration decoder's forward, loss and backward: independent of
                    the seg decoder between the bottleneck and the encoder backward) run there in their own
                    order, their weight gradients inline;
        and sync_op('fork'|'join', lane) entries order a lane against the main stream.  A lane missing from
        `lanes` falls back to the main stream (the list order is a valid sequential schedule).  Works eagerly and
        under hipGraph capture (fork/join through events).  Returns the set of lanes that still have to be joined.
        wrap(op, stream, launch) may time a launch: it must call launch() exactly once."""
        open_lanes = set()
        for op in ops:
            fn, args = op[0], op[1]
            meta = op[2] if len(op) > 2 else None
            if fn is None:
                kind, lane = args
                st = lanes.get(lane)
                if st is None:
                    continue
                if kind == 'fork':
                    st.wait_stream(main)
                    open_lanes.add(lane)
                elif lane in open_lanes:
                    main.wait_stream(st)
                    open_lanes.discard(lane)
                continue
            st = main
            if meta is not None:
                if meta.get('lane') == 'rec' and 'rec' in lanes:
                    st = lanes['rec']
                elif meta.get('side') and 'side0' in lanes:
                    name = 'side%d' % meta.get('side_idx', 0)
                    if name not in lanes:
                        name = 'side0'
                    st = lanes[name]
                    st.wait_stream(main)
                    open_lanes.add(name)

            def launch(fn=fn, args=args, st=st):
                err = fn(*args, st.cuda_stream)
                if err:
                    raise RuntimeError('ramdsir HIP launch failed: %s -> %d' % (fn.__name__, err))
            if wrap is None:
                launch()
            else:
                wrap(op, st, launch)
        return open_lanes


class LaunchList:
    """A launch list compiled for rd_run_list (include/ramdsir.h): the same entries, lanes and fork / join semantics as
    Plan.run_lanes, walked in C++ with ONE ctypes call per segment instead of one call + one event record / stream wait per entry
    (2.2 ms of host time per step -> a fraction).  `lane_names`: the lanes that exist, in stream-index order (index 0 is the main
    stream); entries of a lane that does not exist run on the main stream, exactly as in run_lanes.  The descriptors referenced by
    the entries are owned by the plans (Plan.keep); `ops` is kept so that they stay alive with the list."""

    def __init__(self, ops, lane_names):
        self.lane_names = list(lane_names)
        idx = {name: i + 1 for i, name in enumerate(self.lane_names)}
        entries = []
        for op in ops:
            fn, args = op[0], op[1]
            meta = op[2] if len(op) > 2 else None
            e = L.RdLaunch()
            if fn is None:
                kind, lane = args
                if lane not in idx:
                    continue
                e.op, e.lane = (L.OP_FORK if kind == 'fork' else L.OP_JOIN), idx[lane]
                entries.append(e)
                continue
            e.op = L.OP_CODES[fn.__name__]
            if meta is not None:
                if meta.get('lane') == 'rec' and 'rec' in idx:
                    e.lane = idx['rec']
                elif meta.get('side') and 'side0' in idx:
                    name = 'side%d' % meta.get('side_idx', 0)
                    e.lane, e.wait_main = idx[name if name in idx else 'side0'], 1
            types = fn.argtypes[:-1]                         # without the trailing stream
            assert len(types) == len(args) <= 18, (fn.__name__, len(types), len(args))
            e.nargs = len(args)
            for i, (v, t) in enumerate(zip(args, types)):
                e.a[i] = L.pack_arg(v, t)
            entries.append(e)
        self.n = len(entries)
        self.arr = (L.RdLaunch * max(self.n, 1))(*entries)
        self.ops = ops
        self._bad = C.c_int(-1)

    def run(self, main, lanes, open_mask=0):
        """Enqueue the list; returns the bitmask of lanes left open (bit k = lane_names[k - 1])."""
        streams = (L.vp * (1 + len(self.lane_names)))(main.cuda_stream, *[lanes[nm].cuda_stream for nm in self.lane_names])
        mask = C.c_uint32(open_mask)
        err = L.lib().rd_run_list(self.arr, self.n, streams, len(streams), C.byref(mask), C.byref(self._bad))
        if err:
            bad = self._bad.value
            raise RuntimeError('ramdsir launch list failed at entry %d (op %d): error %d' % (bad, self.arr[bad].op if 0 <= bad < self.n else -1, err))
        return mask.value

    def join(self, main, lanes, mask):
        if mask:
            streams = (L.vp * (1 + len(self.lane_names)))(main.cuda_stream, *[lanes[nm].cuda_stream for nm in self.lane_names])
            L.check(L.lib().rd_join_lanes(streams, len(streams), mask), 'rd_join_lanes')


def sync_op(kind, lane):
    """Pseudo-op for Plan.run_lanes: 'fork' = the lane's stream waits for the main stream here, 'join' = the
    main stream waits for everything the lane has been given so far."""
    assert kind in ('fork', 'join')
    return (None, (kind, lane), {})


def tag_lane(ops, lane):
    """Copy of `ops` with meta['lane'] = lane."""
    out = []
    for op in ops:
        meta = dict(op[2]) if len(op) > 2 else {}
        meta['lane'] = lane
        out.append((op[0], op[1], meta))
    return out


def interleave(a, b):
    """Merge two independent op lists, alternating, each keeping its own order (CPU enqueue order only)."""
    out, i, j = [], 0, 0
    while i < len(a) or j < len(b):
        if i < len(a):
            out.append(a[i]); i += 1
        if j < len(b):
            out.append(b[j]); j += 1
    return out


class WeightPack:
    """Packed (forward and dgrad) copies of every conv weight in `dtype`, refreshed by one launch."""

    def __init__(self, bank, modules, dtype):
        self.bank, self.dtype = bank, dtype
        self.dt = L.RD_BF16 if dtype == torch.bfloat16 else L.RD_F32
        self.ck = 32 if dtype == torch.bfloat16 else 16
        lib = L.lib()
        entries, self.off = [], {}
        total = 0
        for mname, specs in modules:
            for key, shape, kind, _ in specs:
                if kind == 'param' and len(shape) == 4:
                    cout, cin, k, _ = shape
                    taps = k * k
                    for tr in (0, 1):
                        n = lib.rd_packed_elems(cout, cin, taps, tr, self.dt)
                        e = L.RdPackEntry()
                        e.src_off = bank.index[(mname, key)][0]
                        e.dst_off = e.start = total
                        rows, cols = (cin, cout) if tr else (cout, cin)
                        e.Cout, e.Cin, e.taps, e.transpose = cout, cin, taps, tr
                        e.RowPad, e.ColPad = (rows + 31) // 32 * 32, (cols + self.ck - 1) // self.ck * self.ck
                        assert e.RowPad * e.ColPad * taps == n
                        entries.append(e)
                        self.off[(mname, key[:-len('.weight')], bool(tr))] = total
                        total += n
        self.total = total
        self.packed = torch.zeros(total, dtype=dtype, device=bank.device)
        arr = (L.RdPackEntry * len(entries))(*entries)
        raw = bytes(memoryview(arr))
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(bank.device)
        self.n_entries = len(entries)
        self.esize = self.packed.element_size()

    def ptr(self, mname, conv, transpose):
        return self.packed.data_ptr() + self.off[(mname, conv, transpose)] * self.esize

    def pads(self, Cout, Cin):
        return (Cin + self.ck - 1) // self.ck * self.ck, (Cout + 31) // 32 * 32

    def refresh_op(self):
        return (L.lib().rd_pack_weights_batched, (self.bank.params.data_ptr(), self.packed.data_ptr(), self.table.data_ptr(),
                                                  self.n_entries, self.total, self.dt))

    def refresh(self, stream=None):
        fn, args = self.refresh_op()
        L.check(fn(*args, stream), 'pack_weights_batched')


# ------------------------------------------------------------------------------------------------ network builders
def build_convd(plan, x, mode, cout, prefix, mname, norm='bn'):
    """ConvD.forward (unet.py:52-72): [maxpool] conv3-bn / conv3-bn-act / conv3-bn-act.  x: source Act read in `mode`
    (RD_SRC_POOL = the 2x2 max-pool of levels 2-5 fused into the read).  Returns the Act of conv3 (BN3+act pending)."""
    bank = plan.bank
    H, W = (x.H // 2, x.W // 2) if mode == L.SRC_POOL else (x.H, x.W)
    z1 = plan.conv(mname, prefix + 'conv1', [(x, mode, 0, -1)], cout, 9, Norm(bank, mname, prefix + 'bn1', kind=norm), act=False, H=H, W=W)
    z2 = plan.conv(mname, prefix + 'conv2', [(z1, L.SRC_AFF, 0, -1)], cout, 9, Norm(bank, mname, prefix + 'bn2', kind=norm), act=True, H=H, W=W)
    return plan.conv(mname, prefix + 'conv3', [(z2, L.SRC_AFFACT, 0, -1)], cout, 9, Norm(bank, mname, prefix + 'bn3', kind=norm), act=True, H=H, W=W)


def build_encoder(plan, x_act, n=16, mname='enc', norm='bn'):
    """Encoder.forward (unet.py:264-271).  x_act: Act of the image batch (no norm).  Returns [z3_1..z3_5]."""
    feats = []
    prev, prev_mode = x_act, L.SRC_RAW
    for l in range(1, 6):
        z3 = build_convd(plan, prev, prev_mode, n * (1 << (l - 1)), 'convd%d.' % l, mname, norm)
        feats.append(z3)
        prev, prev_mode = z3, L.SRC_POOL
    return feats


def _feat_mode(a):
    return L.SRC_AFFACT if a.norm is not None else L.SRC_RAW


def build_convu(plan, x, skip, planes, first, prefix, mname, norm='bn'):
    """ConvU.forward (unet.py:96-117): [conv3-bn-act] up2 conv1-bn-act cat[skip, .] conv3-bn-act."""
    bank = plan.bank
    if not first:
        x = plan.conv(mname, prefix + 'conv1', [(x, _feat_mode(x), 0, -1)], planes, 9, Norm(bank, mname, prefix + 'bn1', kind=norm), act=True,
                      H=x.H, W=x.W)
    # 1x1 conv below the upsample (commutes with bilinear interpolation); BN2 statistics on up(t)
    t = plan.conv(mname, prefix + 'conv2', [(x, _feat_mode(x), 0, -1)], planes // 2, 1, Norm(bank, mname, prefix + 'bn2', kind=norm), act=True,
                  up_out=True, H=x.H, W=x.W)
    return plan.conv(mname, prefix + 'conv3', [(skip, _feat_mode(skip), 0, -1), (t, L.SRC_UP, 0, -1)], planes, 9,
                     Norm(bank, mname, prefix + 'bn3', kind=norm), act=True, H=skip.H, W=skip.W)


def build_decoder(plan, feats, n=16, num_classes=2, mname='dec', norm='bn'):
    """Decoder.forward (unet.py:290-296).  feats: 5 Acts (raw+pending BN, or materialised RAW inputs)."""
    x = feats[4]
    for l, planes, first, skip in ((4, 16 * n, True, feats[3]), (3, 8 * n, False, feats[2]), (2, 4 * n, False, feats[1]),
                                   (1, 2 * n, False, feats[0])):
        x = build_convu(plan, x, skip, planes, first, 'convu%d.' % l, mname, norm)
    return plan.conv(mname, 'out1', [(x, L.SRC_AFFACT, 0, -1)], num_classes, 9, None, H=x.H, W=x.W)


def build_convu_rec(plan, x, mode, n_off, g_fixed, planes, domains, prefix, mname):
    """ConvU_Rec.forward (unet.py:139-165): conv3-dsbn-act up2 conv1-dsbn-act conv3-dsbn-act; group g uses bns[domains[g]]."""
    bank = plan.bank
    h = planes // 2
    nrm = (lambda nm: Norm(bank, mname, nm, domains)) if domains is not None else (lambda nm: Norm(bank, mname, nm))
    z1 = plan.conv(mname, prefix + 'conv1', [(x, mode, n_off, g_fixed)], h, 9, nrm(prefix + 'bn1'), act=True, H=x.H, W=x.W)
    t = plan.conv(mname, prefix + 'conv2', [(z1, L.SRC_AFFACT, 0, -1)], h, 1, nrm(prefix + 'bn2'), act=True, up_out=True, H=x.H, W=x.W)
    return plan.conv(mname, prefix + 'conv3', [(t, L.SRC_UP, 0, -1)], h, 9, nrm(prefix + 'bn3'), act=True, H=2 * x.H, W=2 * x.W)


def build_rec_decoder(plan, x5, n_off, g_fixed, domains, n=16, num_classes=3, mname='rec'):
    """Rec_Decoder.forward (unet.py:316-322) for all domain slices at once; plan.G groups, group g uses
    DomainSpecificBatchNorm2d.bns[domains[g]] (dsbn.py:26).  x5: bottleneck Act (of another plan when
    n_off/g_fixed address a slice of the encoder batch)."""
    x, mode = x5, _feat_mode(x5)
    for l, planes in ((4, 16 * n), (3, 8 * n), (2, 4 * n), (1, 2 * n)):
        x = build_convu_rec(plan, x, mode, n_off if x is x5 else 0, g_fixed if x is x5 else -1, planes, domains, 'convu%d.' % l, mname)
        mode = L.SRC_AFFACT
    return plan.conv(mname, 'out1', [(x, L.SRC_AFFACT, 0, -1)], num_classes, 9, None, H=x.H, W=x.W)
