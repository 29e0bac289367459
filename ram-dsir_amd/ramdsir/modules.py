"""nn.Module machinery behind the drop-in ``networks.unet`` classes: parameters are nn.Parameters that alias
a ParamBank arena, forward/backward of a whole Encoder / Decoder / Rec_Decoder are ONE autograd.Function
each that runs the HIP launch lists of an engine.Plan (there is no ATen compute and no CPU fallback)."""
import ctypes as C
import os
import weakref

import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E


_storage_override = None


def set_storage_dtype(dtype):
    """Activation storage type of the module-level path for this process (torch.float32 / torch.bfloat16; None: back to the
    RAMDSIR_DTYPE environment default).  ramdsir.trainer.ModuleTrainer sets it from its `dtype` argument, so that
    `train.py --norm gn --dtype bf16` really trains in bf16 storage."""
    global _storage_override
    if dtype not in (None, torch.float32, torch.bfloat16):
        raise ValueError('storage dtype must be torch.float32 or torch.bfloat16, got %r' % (dtype,))
    _storage_override = dtype


def storage_dtype():
    """Activation storage type of the module-level path: set_storage_dtype() if called, else fp32 (default, parity-grade) or bf16
    (RAMDSIR_DTYPE=bf16).  The fused trainer chooses its own dtype."""
    if _storage_override is not None:
        return _storage_override
    return torch.bfloat16 if os.environ.get('RAMDSIR_DTYPE', 'f32') == 'bf16' else torch.float32


class FusedConv2d(nn.Conv2d):
    """Holds weight/bias exactly like nn.Conv2d (same state_dict keys, isinstance checks and init loops of
    unet.py:257-262 work); the arithmetic runs inside the owning module's fused launch list."""

    def forward(self, x):
        raise NotImplementedError('this conv is executed inside the fused HIP graph of its parent module')


class FusedBatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d whose arithmetic is HIP (``isinstance(m, nn.BatchNorm2d)`` and ``m.train()`` of
    test_fundus_slice.py:75-83 work).  Inside Encoder / Decoder / Rec_Decoder and their blocks the statistics and
    the affine run fused in the conv kernels; called on its own (as DomainSpecificBatchNorm2d.forward does,
    dsbn.py:24-27) it runs rd_bn_stats -> rd_bn_finalize_fwd -> rd_nhwc_to_nchw, and backward through rd_grad_in ->
    rd_bn_finalize_bwd -> rd_bn_apply."""

    def forward(self, x):
        self._check_input_dim(x)
        if not (x.is_cuda and self.affine and self.track_running_stats and self.momentum is not None):
            raise RuntimeError('FusedBatchNorm2d: CUDA input, affine=True, track_running_stats=True, momentum set '
                               '(the only configuration the reference builds, unet.py:19 / dsbn.py:10-11); no CPU fallback')
        return BatchNormFn.apply(x, self.weight, self.bias, self)


class FusedGroupNorm(nn.GroupNorm):
    """The nn.GroupNorm(1, planes) slot of normalization(planes, 'gn') (code/networks/unet.py:20-21): holds weight / bias under the
    reference's state_dict keys; the arithmetic runs inside the launch list of the parent block (one statistics group per image,
    rd_gn_finalize_fwd / _bwd)."""

    def forward(self, x):
        raise NotImplementedError('this GroupNorm is executed inside the fused HIP graph of its parent module')


class FusedInstanceNorm2d(nn.InstanceNorm2d):
    """The nn.InstanceNorm2d(planes) slot of normalization(planes, 'in') (unet.py:22-23; affine=False, no running statistics: no
    state_dict entries); runs inside the parent block's launch list as BatchNorm statistics with one group per image."""

    def forward(self, x):
        raise NotImplementedError('this InstanceNorm is executed inside the fused HIP graph of its parent module')


def group_starts(norm, N):
    """gstart of a module-level launch plan: one BatchNorm group for the whole batch, one group per image for gn / in."""
    if norm in ('gn', 'in'):
        if N > L.MAXG:
            raise ValueError('norm=%r keeps one statistics group per image: at most %d images per call, got %d' % (norm, L.MAXG, N))
        return list(range(N + 1))
    return [0, N]


class BatchNormFn(torch.autograd.Function):
    """Standalone BatchNorm2d forward/backward on the HIP kernels (one group: the whole batch shares the statistics)."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn):
        lib = L.lib()
        N, Cc, H, W = x.shape
        dev, st = x.device, _stream()
        dtype = storage_dtype()
        dt = L.RD_BF16 if dtype == torch.bfloat16 else L.RD_F32
        for t in (weight, bias, bn.running_mean, bn.running_var):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise RuntimeError('FusedBatchNorm2d: parameters and buffers must be contiguous fp32 CUDA tensors')
        z = torch.empty(N, H, W, Cc, dtype=dtype, device=dev)
        gs = L.gstart_array([0, N])
        L.check(lib.rd_nchw_to_nhwc(x.contiguous().float().data_ptr(), z.data_ptr(), N, Cc, H, W, Cc, dt, st), 'bn: nchw_to_nhwc')
        coef = torch.empty(4, Cc, dtype=torch.float32, device=dev)            # scale, shift, mean, invstd
        b = L.RdBnFwd()
        stats = None
        if bn.training:
            stats = torch.zeros(L.STAT_SLOTS * Cc * 2, dtype=torch.float64, device=dev)
            L.check(lib.rd_bn_stats(z.data_ptr(), stats.data_ptr(), N, H, W, Cc, 1, gs, dt, st), 'bn_stats')
            b.stats = stats.data_ptr()
        b.conv_bias = None
        b.scale, b.shift, b.mean, b.invstd = (coef[i].data_ptr() for i in range(4))
        b.gamma[0], b.beta[0] = weight.data_ptr(), bias.data_ptr()
        b.running_mean[0], b.running_var[0] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        b.num_batches_tracked[0] = bn.num_batches_tracked.data_ptr()
        b.count[0] = float(N * H * W)
        b.C, b.G, b.eps, b.momentum, b.training = Cc, 1, bn.eps, bn.momentum, 1 if bn.training else 0
        L.check(lib.rd_bn_finalize_fwd(C.byref(b), st), 'bn_finalize_fwd')
        y = torch.empty(N, Cc, H, W, dtype=torch.float32, device=dev)
        L.check(lib.rd_nhwc_to_nchw(z.data_ptr(), y.data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(), 0, 0.0, N, Cc, H, W, 1, gs, dt, st),
                'bn: nhwc_to_nchw')
        ctx.save_for_backward(z, coef, weight)
        ctx.training, ctx.dims, ctx.dt, ctx.keep = bn.training, (N, Cc, H, W), dt, (stats, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError('backward through BatchNorm in eval mode is not implemented in the HIP path')
        lib = L.lib()
        z, coef, weight = ctx.saved_tensors
        N, Cc, H, W = ctx.dims
        dev, st, dt = dy.device, _stream(), ctx.dt
        gs = L.gstart_array([0, N])
        g = torch.empty_like(z)
        bst = torch.zeros(L.STAT_SLOTS * Cc * 2, dtype=torch.float64, device=dev)
        L.check(lib.rd_grad_in(dy.contiguous().float().data_ptr(), z.data_ptr(), g.data_ptr(), None, None, bst.data_ptr(), 0, 0.0, 0,
                               N, Cc, H, W, 1, gs, dt, st), 'bn: grad_in')
        pqr = torch.empty(3, Cc, dtype=torch.float32, device=dev)
        dgb = torch.zeros(2, Cc, dtype=torch.float32, device=dev)
        q = L.RdBnBwd()
        q.bstats, q.mean, q.invstd = bst.data_ptr(), coef[2].data_ptr(), coef[3].data_ptr()
        q.P, q.Q, q.R = (pqr[i].data_ptr() for i in range(3))
        q.gamma[0], q.dgamma[0], q.dbeta[0] = weight.data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr()
        q.count[0] = float(N * H * W)
        q.C, q.G = Cc, 1
        L.check(lib.rd_bn_finalize_bwd(C.byref(q), st), 'bn_finalize_bwd')
        dz = torch.empty_like(z)
        L.check(lib.rd_bn_apply(g.data_ptr(), z.data_ptr(), dz.data_ptr(), pqr[0].data_ptr(), pqr[1].data_ptr(), pqr[2].data_ptr(), 1.0,
                                N, H, W, Cc, 1, gs, dt, st), 'bn: apply bwd')
        dx = torch.empty(N, Cc, H, W, dtype=torch.float32, device=dev)
        L.check(lib.rd_nhwc_to_nchw(dz.data_ptr(), dx.data_ptr(), None, None, 0, 0.0, N, Cc, H, W, 1, gs, dt, st), 'bn: dx')
        return dx, dgb[0], dgb[1], None


def activation_slope(activation):
    return 0.0 if activation == 'relu' else 0.01      # unet.py:47-50: anything but 'relu' is LeakyReLU(0.01)


class FusedModule(nn.Module):
    """Common part of Encoder / Decoder / Rec_Decoder."""
    _mname = 'mod'

    def _finish_init(self, specs, activation, init=True):
        """init=False: a block (ConvD / ConvU / ConvU_Rec) -- the reference initialises only in the constructors of
        Encoder / Decoder / Rec_Decoder (unet.py:257-262), so a block draws nothing beyond nn.Conv2d's own default."""
        self._specs = specs
        self._slope = activation_slope(activation)
        self._plans = {}
        self._bank = None
        self._wpack = None
        self._bound_device = None
        if not init:
            return
        # reference init (unet.py:257-262 and twins)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu' if activation == 'relu' else 'leaky_relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    # ---- parameter arena binding
    def _named_tensors(self):
        d = dict(self.named_parameters())
        d.update(dict(self.named_buffers()))
        return d

    def bind(self, bank, mname, wpack=None):
        """Re-home every parameter / buffer into `bank` (values are kept); used by the fused trainer to put
        the three modules into one arena."""
        cur = self._named_tensors()
        for key, shape, kind, dt in self._specs:
            dst = bank.p(mname, key) if kind == 'param' else bank.b(mname, key)
            dst.copy_(cur[key].detach().to(dst.device))
            cur[key].data = dst
        self._bank, self._mname, self._wpack = bank, mname, wpack
        self._bound_device = bank.device
        self._plans = {}

    def _ensure_bound(self, device):
        if self._bank is None or self._bound_device != device:
            bank = E.ParamBank([(self._mname, self._specs)], device)
            self.bind(bank, self._mname)
        else:
            # .to()/.cuda()/load_state_dict keep .data aliasing unless a tensor was replaced: re-alias if needed
            cur = self._named_tensors()
            for key, shape, kind, dt in self._specs:
                dst = self._bank.p(self._mname, key) if kind == 'param' else self._bank.b(self._mname, key)
                if cur[key].data_ptr() != dst.data_ptr():
                    dst.copy_(cur[key].detach().to(dst.device))
                    cur[key].data = dst
        if self._wpack is None or self._wpack.dtype != storage_dtype():
            self._wpack = E.WeightPack(self._bank, [(self._mname, self._specs)], storage_dtype())
            self._plans = {}

    def _bn_training(self):
        flags = [m.training for m in self.modules() if isinstance(m, (nn.BatchNorm2d, nn.GroupNorm, nn.InstanceNorm2d))]
        return any(flags)

    def _acquire_plan(self, key, builder):
        pool = self._plans.setdefault(key, [])
        for pl in pool:
            if not pl.busy:
                pl.busy = True
                return pl
        pl = builder()
        pl.busy = True
        pool.append(pl)
        return pl

    @staticmethod
    def _release(pl):
        pl.busy = False

    def _param_list(self):
        return [p for _, p in self.named_parameters()]


def _stream():
    return torch.cuda.current_stream().cuda_stream


def to_nhwc(x, act_buf, dt):
    N, Cc, H, W = x.shape
    x = x.contiguous().float()
    L.check(L.lib().rd_nchw_to_nhwc(x.data_ptr(), act_buf.data_ptr(), N, Cc, H, W, Cc, dt, _stream()), 'nchw_to_nhwc')


def materialize(a, plan):
    """NCHW fp32 tensor of act(BN(z)) for a plan Act (what the reference module returns)."""
    out = torch.empty(a.N, a.C, a.H, a.W, dtype=torch.float32, device=a.buf.device)
    sc = a.scale.data_ptr() if a.norm is not None else None
    sh = a.shift.data_ptr() if a.norm is not None else None
    L.check(L.lib().rd_nhwc_to_nchw(a.buf.data_ptr(), out.data_ptr(), sc, sh, 1 if a.act else 0, plan.slope, a.N, a.C, a.H, a.W,
                                    plan.G, plan.gs_arr, plan.dt, _stream()), 'nhwc_to_nchw')
    return out


def grad_in(a, plan, dy):
    """Gradient arriving from torch for a materialised output: mask by the activation, accumulate BN sums."""
    dy = dy.contiguous().float()
    g = a.grad_buf()
    has_bn = a.norm is not None
    L.check(L.lib().rd_grad_in(dy.data_ptr(), a.buf.data_ptr() if has_bn else None, g.data_ptr(),
                               a.scale.data_ptr() if has_bn else None, a.shift.data_ptr() if has_bn else None,
                               a.plan.stat_ptr(a.bstats) if has_bn else None, 1 if (a.act and has_bn) else 0, plan.slope,
                               1 if a.g_written_rt else 0, a.N, a.C, a.H, a.W, plan.G, plan.gs_arr, plan.dt, _stream()), 'grad_in')
    a.g_written_rt = True


def grad_out(a, plan):
    """NCHW fp32 gradient w.r.t. a RAW input Act (a feature tensor handed in by torch)."""
    out = torch.empty(a.N, a.C, a.H, a.W, dtype=torch.float32, device=a.buf.device)
    L.check(L.lib().rd_nhwc_to_nchw(a.grad_buf().data_ptr(), out.data_ptr(), None, None, 0, 0.0, a.N, a.C, a.H, a.W, plan.G, plan.gs_arr,
                                    plan.dt, _stream()), 'grad nhwc_to_nchw')
    return out


class FusedFn(torch.autograd.Function):
    """forward(module, plan, inputs (Acts to fill), outputs (Acts to materialise), n_in, *tensors)"""

    @staticmethod
    def forward(ctx, module, plan, in_acts, out_acts, *tensors):
        n_in = len(in_acts)
        xs = tensors[:n_in]
        for a, x in zip(in_acts, xs):
            to_nhwc(x, a.buf, plan.dt)
        module._wpack.refresh(_stream())
        plan.stat_arena.zero_()
        E.Plan.run(plan.fwd, _stream())
        outs = tuple(materialize(a, plan) for a in out_acts)
        ctx.module, ctx.plan, ctx.in_acts, ctx.out_acts = module, plan, in_acts, out_acts
        ctx.needs_in = [x.requires_grad for x in xs]
        try:
            weakref.finalize(ctx, FusedModule._release, plan)      # forward without backward: free the plan with the graph
        except TypeError:
            pass
        return outs

    @staticmethod
    def backward(ctx, *grads):
        module, plan = ctx.module, ctx.plan
        if not plan.training:
            raise RuntimeError('backward through BatchNorm in eval mode is not implemented in the HIP path')
        bank = module._bank
        lo, hi = bank.module_range[module._mname]
        bank.grads[lo:hi].zero_()
        for a in ctx.out_acts:
            a.g_written_rt = False
        for a, dy in zip(ctx.out_acts, grads):
            if dy is not None:
                grad_in(a, plan, dy)
        for a in ctx.out_acts:
            if not a.g_written_rt:                      # output unused downstream: its gradient is zero
                a.grad_buf().zero_()
        E.Plan.run(plan.bwd, _stream())
        gin = [grad_out(a, plan) if need else None for a, need in zip(ctx.in_acts, ctx.needs_in)]
        gpar = [bank.g(module._mname, key).clone() for key, shape, kind, dt in module._specs if kind == 'param']
        FusedModule._release(plan)
        return (None, None, None, None) + tuple(gin) + tuple(gpar)


def run_fused(module, plan, in_acts, out_acts, xs):
    params = module._param_list()
    if torch.is_grad_enabled() and (any(p.requires_grad for p in params) or any(x.requires_grad for x in xs)):
        return FusedFn.apply(module, plan, in_acts, out_acts, *xs, *params)
    with torch.no_grad():
        try:
            for a, x in zip(in_acts, xs):
                to_nhwc(x, a.buf, plan.dt)
            module._wpack.refresh(_stream())
            plan.stat_arena.zero_()
            E.Plan.run(plan.fwd, _stream())
            return tuple(materialize(a, plan) for a in out_acts)
        finally:
            FusedModule._release(plan)
