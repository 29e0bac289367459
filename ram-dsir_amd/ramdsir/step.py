"""The fused RAM-DSIR training step (code/train.py:225-296 fundus, :393-465 prostate) as one static
launch list on one HIP stream, optionally captured into a hipGraph:

    [RAM mix] -> encoder+seg-decoder on [img ; img_freq] (2B images, 2 BN groups)
              -> restoration decoder on the img_freq bottleneck (B images, one DSBN group per domain)
              -> fused seg/consistency loss + rec loss (dlogits)
              -> seg-decoder backward -> rec-decoder backward -> encoder backward
              -> Adam (3 param groups, poly LR) -> repack conv weights

Only the flag combination that runs in the reference (--ram --rec [--consistency]) exists here.
"""
import ctypes as C
import os

import torch

from . import _lib as L
from . import engine as E
from . import ram as R
from . import streams
from . import tuning as T


class TrainStep:
    def __init__(self, bank, mods, dtype, batch_sizes, H, W, dataset='fundus', consistency='kd', lambda_rec=0.1, lr=2e-3,
                 total_iters=21200, in_channels=3, n=16, num_classes=2, slope=0.0, wpack=None, ram=False, options=None):
        self.bank, self.dtype = bank, dtype
        self.dt = L.RD_BF16 if dtype == torch.bfloat16 else L.RD_F32
        self.batch_sizes = list(batch_sizes)
        self.B = B = sum(batch_sizes)
        self.H, self.W, self.K, self.c = H, W, num_classes, in_channels
        self.dataset = dataset
        dev = bank.device
        lib = L.lib()
        self.wpack = wpack if wpack is not None else E.WeightPack(bank, mods, dtype)
        # ---- graphs
        self.opt = opt = T.options(options)
        self.seg = E.Plan(bank, dtype, 2 * B, [0, B, 2 * B], slope=slope)
        self.seg.pad_narrow = self.seg.materialize_up = True
        self.seg.materialize_pool = bool(opt['pool_mat'])
        self.seg.fused_bwd = bool(opt['fused_bwd'])
        # lane budgets are tuned for the bf16 kernels (in fp32 the weight gradients are several times heavier and the side lane
        # itself becomes the critical path when it is narrowed: 315 -> 268 images/s)
        budget = bool(opt['fork']) and dtype == torch.bfloat16
        self.seg.side_cus = T.cu_budget(opt['side_cus'], dev) if budget else 0
        self.seg.materialize_min_c = opt['mat_min_c'] if opt['mat_min_c'] > 0 else None
        self.seg.materialize_dz_min_c = opt['mat_dz_min_c'] if opt['mat_dz_min_c'] > 0 else None
        slot = self.seg.slot_channels()
        self.x = E.Act(self.seg, 2 * B, H, W, in_channels, name='input', cstride=slot if in_channels < slot else None)
        self.feats = E.build_encoder(self.seg, self.x, n=n)
        self.logits = E.build_decoder(self.seg, self.feats, n=n, num_classes=num_classes)
        gs = [0]
        for b in batch_sizes:
            gs.append(gs[-1] + b)
        self.rec = E.Plan(bank, dtype, B, gs, slope=slope)
        self.rec.pad_narrow = self.rec.materialize_up = True
        self.rec.materialize_pool = self.seg.materialize_pool
        self.rec.fused_bwd = self.seg.fused_bwd
        lane = bool(budget and opt['rec_lane'])
        self.rec.side_cus = T.cu_budget(opt['rec_cus'], dev) if lane else 0    # its weight gradients run inline on its own lane
        self.rec.conv_cus = T.cu_budget(opt['rec_cus'], dev) if lane else 0
        self.rec.materialize_min_c = self.seg.materialize_min_c
        self.rec.materialize_dz_min_c = self.seg.materialize_dz_min_c
        self.rec_logits = E.build_rec_decoder(self.rec, self.feats[4], n_off=B, g_fixed=1, domains=list(range(len(batch_sizes))),
                                              n=n, num_classes=in_channels)
        self.seg.build(self.wpack)
        self.rec.build(self.wpack)
        # weight-gradient split workspaces: one per plan, because the restoration decoder's backward runs beside
        # the seg decoder's on its own stream
        self.n_side = max(1, int(opt['side_streams']))
        self.ws = [E.workspace(max(self.seg.ws_bytes, 4) // 4 + 1, dev) for _ in range(self.n_side)]
        self.rec_wsp = E.workspace(max(self.rec.ws_bytes, 4) // 4 + 1, dev)
        self.seg.bind_workspace(self.ws)
        self.rec.bind_workspace(self.rec_wsp)
        # ---- losses
        if dataset == 'fundus':
            self.target = torch.zeros(B, num_classes, H, W, dtype=torch.float32, device=dev)
        else:
            self.target = torch.zeros(B, H, W, dtype=torch.int64, device=dev)
        self.losses = torch.zeros(8, dtype=torch.float32, device=dev)
        self.rec_mse = torch.zeros(len(batch_sizes), dtype=torch.float32, device=dev)
        sl = L.RdSegLoss()
        sl.logits, sl.target = self.logits.buf.data_ptr(), self.target.data_ptr()
        sl.dlogits, sl.losses_out = self.logits.grad_buf().data_ptr(), self.losses.data_ptr()
        sl.B, sl.H, sl.W, sl.K = B, H, W, num_classes
        sl.dlogits_cstride = self.logits.gCs
        sl.kind = 0 if dataset == 'fundus' else 1
        sl.consistency = {None: 0, 'kd': 1, 'mse': 2}[consistency]
        sl.cons_weight = 0.5
        self.seg_ws = E.workspace(lib.rd_seg_loss_workspace(C.byref(sl)) // 4, dev)
        sl.partial = self.seg_ws.data_ptr()
        self.sl = sl
        self.rec_ws = E.workspace(lib.rd_rec_loss_workspace(B, H, W, in_channels) // 4, dev)
        self.lambda_rec = lambda_rec
        # ---- optimizer
        bank.ensure_adam()
        self.iter = torch.zeros((), dtype=torch.int32, device=dev)
        self.hyper = torch.zeros(4, dtype=torch.float32, device=dev)
        ad = L.RdAdam()
        ad.param, ad.grad = bank.params.data_ptr(), bank.grads.data_ptr()
        ad.exp_avg, ad.exp_avg_sq = bank.exp_avg.data_ptr(), bank.exp_avg_sq.data_ptr()
        ad.n, ad.n_half_lr = bank.n, bank.module_range['enc'][1]
        ad.iter, ad.hyper_out = self.iter.data_ptr(), self.hyper.data_ptr()
        ad.base_lr, ad.total_iters, ad.beta1, ad.beta2, ad.eps = lr, total_iters, 0.9, 0.999, 1e-8
        self.ad = ad
        # ---- RAM (optional: raw images + partner images + lambda in, both network inputs out)
        # ram = True: fp32 source / partner buffers; ram = 'u8': uint8 buffers (decoded PNG pixels of the Fundus pipeline:
        # 1 byte per value over PCIe and out of HBM; the values are the same integers the reference holds as float32)
        self.ram = None
        if ram:
            idt = torch.uint8 if ram == 'u8' else torch.float32
            self.src = torch.zeros(B, H, W, in_channels, dtype=idt, device=dev)
            self.trg = torch.zeros(B, H, W, in_channels, dtype=idt, device=dev)
            self.lam = torch.ones(B, dtype=torch.float32, device=dev)
            self.ram = R.RamMixer(B, H, W, dtype, dev, dataset)
            self.ram.bind(self.src, self.trg, self.lam, self.x.buf[:B], self.x.buf[B:])
        self.graph = None
        self._zero_args = None
        # eager execution uses three HIP streams (Plan.run_lanes): main = the forward / dgrad chain, 'side' = the
        # weight-gradient kernels beside it, 'rec' = the whole restoration-decoder branch beside the seg decoder.
        # RD_FORK=0: everything on one stream.  Captured into a hipGraph the same forks become parallel branches,
        # which ROCm 7's graph executor runs SLOWER than the single chain (measured), so
        # capture() records the one-stream order.
        # The lanes must sit on different HARDWARE queues than the caller's stream and than each other: streams.pick_lanes
        # measures that instead of trusting the creation order (a process group or a DataLoader created before this object
        # shifts the stream -> queue mapping: 5.7 -> 6.5 ... 11 ms/step when two lanes share a queue)
        with torch.cuda.device(dev):
            picked = streams.pick_lanes(self.n_side + 1, dev, [torch.cuda.current_stream(dev)]) if opt['fork'] else \
                [torch.cuda.Stream(device=dev) for _ in range(self.n_side + 1)]
        self.side, self.rec_stream = picked[:self.n_side], picked[self.n_side]
        # True: every lane stream was MEASURED to run beside the main stream and the other lanes (bench.py reports it)
        self.lanes_verified = bool(opt['fork']) and all(streams.verified.get(id(st), False) for st in picked)
        self.fork = bool(opt['fork'])
        self.rec_lane = bool(opt['rec_lane'])
        self._ops = self._build_ops()

    def _build_ops(self):
        """Segments: A = [RAM] forward, losses, seg-decoder + rec-decoder backward; B = encoder backward (B1 = levels
        5..3, B2 = levels 2..1); C = Adam + weight repack.  Single GPU runs A+B+C back to back; data parallel
        all-reduces each part of the flat gradient as soon as the segment that completes it ends (ddp.py).
        Inside A the restoration decoder (forward, loss, backward up to its last dgrad) is tagged lane='rec': it only
        meets the seg decoder again in the bottleneck gradient, which its last dgrad ACCUMULATES into after the
        seg decoder's has written it -- that launch stays on the main stream behind the join."""
        lib = L.lib()
        B, H, W = self.B, self.H, self.W
        split = self.seg.bwd_split['enc']
        fdec = self.seg.fwd_split['dec']
        seg_loss = (lib.rd_seg_loss, (C.byref(self.sl), self.dt))
        rec_loss = (lib.rd_rec_loss, (self.rec_logits.buf.data_ptr(), self.x.buf.data_ptr(), self.rec_logits.grad_buf().data_ptr(),
                                      self.rec_mse.data_ptr(), self.rec_ws.data_ptr(), B, H, W, self.c, self.x.Cs, self.rec_logits.gCs,
                                      self.rec.G, self.rec.gs_arr,
                                      self.lambda_rec, self.dt))
        a = []
        if self.ram is not None:
            a.append(self.ram.op())
        a += self.seg.fwd[:fdec]                                     # encoder
        a.append(E.sync_op('fork', 'rec'))
        rec_branch = E.tag_lane(self.rec.fwd + [rec_loss] + self.rec.bwd[:-1], 'rec')
        dec_branch = self.seg.fwd[fdec:] + [seg_loss] + self.seg.bwd[:split]
        a += E.interleave(dec_branch, rec_branch)
        a.append(E.sync_op('join', 'rec'))
        a.append(self.rec.bwd[-1])                                   # rec.convu4.conv1 dgrad: += into the bottleneck gradient
        assert self.rec.bwd[-1][2].get('what') == 'dgrad', self.rec.bwd[-1][2]
        b = list(self.seg.bwd[split:])
        # encoder backward in two parts (data parallel: the deep levels hold 98 % of the encoder's parameters and finish
        # first, so their gradients are all-reduced while the 200x200 / 400x400 levels are still running)
        cut = self.seg.bwd_node_start.get(('enc', 'convd2.conv3'), split) - split
        self.seg_b1, self.seg_b2 = b[:cut], b[cut:]
        self.enc_deep_offset = min(off for (m, k), (off, _) in self.bank.index.items() if m == 'enc' and k.startswith('convd3.'))
        c = [(lib.rd_adam_step, (C.byref(self.ad),)), self.wpack.refresh_op()]
        self.seg_a, self.seg_b, self.seg_c = a, b, c
        return a + b + c

    def lanes(self):
        if not self.fork:
            return {}
        out = {'side%d' % k: st for k, st in enumerate(self.side)}
        if self.rec_lane:
            out['rec'] = self.rec_stream
        return out

    # ---- inputs
    def load_images(self, img_nchw, img_freq_nchw, stream=None):
        """fp32 NCHW device tensors in [-1,1] (what the reference's DataLoader yields, train.py:244)."""
        lib, B = L.lib(), self.B
        s = self._stream() if stream is None else stream
        half = self.x.buf[B:]
        L.check(lib.rd_nchw_to_nhwc(img_nchw.data_ptr(), self.x.buf.data_ptr(), B, self.c, self.H, self.W, self.x.Cs, self.dt, s), 'load img')
        L.check(lib.rd_nchw_to_nhwc(img_freq_nchw.data_ptr(), half.data_ptr(), B, self.c, self.H, self.W, self.x.Cs, self.dt, s), 'load img_freq')

    def load_raw(self, src_nhwc, trg_nhwc, lam):
        """RAM inputs: what Fundus_Multi.__getitem__ holds before the FFTs (fundus.py:209-212): the
        transformed image and the partner image as HWC arrays, and the mix ratio.  The buffers are uint8 when the step
        was built with ram='u8' (decoded PNG pixels) and float32 otherwise; a float image handed to a uint8 step would be
        truncated / wrapped by the copy where the reference mixes it as float32, so that combination is refused."""
        for name, t in (('src', src_nhwc), ('trg', trg_nhwc)):
            if self.src.dtype == torch.uint8 and t.dtype != torch.uint8:
                raise TypeError('%s images are %s but this TrainStep was built with ram=\'u8\' (uint8 buffers): pass uint8 '
                                'pixels or build the step with ram=True (float32 buffers)' % (name, t.dtype))
        self.src.copy_(src_nhwc)
        self.trg.copy_(trg_nhwc)
        self.lam.copy_(lam)

    def load_target(self, mask):
        self.target.copy_(mask)

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    # ---- execution
    def zero(self, stream=None):
        """optimizer.zero_grad() (train.py:285) + the BatchNorm sum arenas, through the library (rd_zero)."""
        if self._zero_args is None:
            ts = (self.seg.stat_arena, self.rec.stat_arena, self.bank.grads)
            self._zero_args = ((L.vp * 3)(*[t.data_ptr() for t in ts]), (L.i64 * 3)(*[t.numel() * t.element_size() for t in ts]))
        L.check(L.lib().rd_zero(self._zero_args[0], self._zero_args[1], 3, self._stream() if stream is None else stream), 'rd_zero')

    def run_segment(self, ops, main=None, lanes=None, wrap=None):
        """One segment over the lanes; every lane it used is joined before returning."""
        main = torch.cuda.current_stream() if main is None else main
        lanes = self.lanes() if lanes is None else lanes
        for lane in E.Plan.run_lanes(ops, main, lanes, wrap):
            main.wait_stream(lanes[lane])

    def run_eager(self, lanes=None):
        self.zero()
        self.run_segment(self.seg_a + self.seg_b, lanes=lanes)   # one join of the weight-gradient stream, right before Adam
        self.run_segment(self.seg_c, lanes=lanes)

    def capture(self):
        """Capture one step (zeroing + every launch) into a hipGraph on a side stream."""
        self._refuse_budgets_on_one_chain('capture()')
        self.wpack.refresh(self._stream())
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            # warm-up outside capture so that lazy module loading / attribute setting is done
            saved = self._snapshot()
            self.run_eager()
            st.synchronize()
            self._restore(saved)
            with torch.cuda.graph(g, stream=st):
                self.run_eager(lanes=self.lanes() if self.opt['graph_fork'] else {})
        torch.cuda.current_stream().wait_stream(st)
        self.graph = g
        return g

    def _refuse_budgets_on_one_chain(self, what):
        """The lane budgets (tuning.py side_cus / rec_cus) are baked into the launch descriptors and the weight-gradient
        workspace when the plans are built; on ONE chain (a captured graph without forked branches) they would silently run
        the weight gradients and the restoration branch on half of the GPU."""
        if (self.seg.side_cus or self.rec.side_cus or self.rec.conv_cus) and not self.opt['graph_fork']:
            raise ValueError('%s runs the step as one chain, but this TrainStep was built with lane budgets (side_cus=%d, rec_cus=%d): '
                             'build it with options=dict(side_cus=0, rec_cus=0)' % (what, self.seg.side_cus, self.rec.conv_cus))

    def _snapshot(self):
        b = self.bank
        return (b.params.clone(), b.exp_avg.clone(), b.exp_avg_sq.clone(), {k: v.clone() for k, v in b.buffers.items()},
                self.iter.clone())

    def _restore(self, saved):
        b = self.bank
        b.params.copy_(saved[0]); b.exp_avg.copy_(saved[1]); b.exp_avg_sq.copy_(saved[2])
        for k, v in saved[3].items():
            b.buffers[k].copy_(v)
        self.iter.copy_(saved[4])
        self.wpack.refresh(self._stream())

    def step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.run_eager()

    def loss_dict(self):
        """Synchronises.  Names follow the tensorboard scalars of train.py:298-304."""
        l = self.losses.cpu().tolist()
        r = self.rec_mse.cpu().tolist()
        seg = 'bce' if self.dataset == 'fundus' else 'ce'
        return {'loss_%s_1' % seg: l[0], 'loss_dice_1': l[1], 'loss_%s_2' % seg: l[2], 'loss_dice_2': l[3],
                'loss_consistency': l[4], 'rec': r, 'loss': l[5] + self.lambda_rec * sum(r)}


def make_bank(device, in_channels=3, n=16, num_classes=2, num_domains=3):
    mods = [('enc', E.encoder_specs(in_channels, n)), ('dec', E.decoder_specs(n, num_classes)),
            ('rec', E.rec_decoder_specs(n, in_channels, num_domains))]
    return E.ParamBank(mods, device), mods


def load_state(bank, mname, sd):
    """Copy a reference-layout state_dict (any device) into the bank's arenas / buffers."""
    for k, v in sd.items():
        if (mname, k) in bank.index:
            bank.p(mname, k).copy_(v.to(bank.device))
        elif (mname, k) in bank.buffers:
            bank.buffers[(mname, k)].copy_(v.to(bank.device))
        else:
            raise KeyError('unexpected key %s.%s' % (mname, k))


def state_dict_of(bank, mname, specs):
    from collections import OrderedDict
    out = OrderedDict()
    for key, shape, kind, _ in specs:
        out[key] = (bank.p(mname, key) if kind == 'param' else bank.b(mname, key)).detach().clone()
    return out
