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
        self.seg.pad_narrow = True
        self.seg.materialize_up = bool(opt['up_mat'])
        self.seg.materialize_pool = bool(opt['pool_mat'])
        self.seg.fused_bwd = bool(opt['fused_bwd'])
        self.seg.fold_finalize = int(opt['fold_finalize'])
        self.seg.fold_wgrad_behind = bool(opt['fold_wgrad_behind'])
        self.seg.fold_fwd_kinds = int(opt['fold_fwd_kinds'])
        self.seg.store_operands = int(opt['store_wgrad_operands'])
        self.seg.store_min_c = int(opt['store_wgrad_min_c'])
        # lane budgets are tuned for the bf16 kernels (in fp32 the weight gradients are several times heavier and the side lane
        # itself becomes the critical path when it is narrowed: 315 -> 268 images/s)
        budget = bool(opt['fork']) and dtype == torch.bfloat16
        self.seg.side_cus = T.cu_budget(opt['side_cus'], dev) if budget else 0
        self.seg.dgrad_cus = T.cu_budget(opt['dgrad_cus'], dev) if budget else 0
        self.seg.materialize_min_c = opt['mat_min_c'] if opt['mat_min_c'] > 0 else None
        self.seg.materialize_dz_min_c = opt['mat_dz_min_c'] if opt['mat_dz_min_c'] > 0 else None
        self.seg.materialize_dz_wide = bool(opt['mat_dz_wide'])
        slot = self.seg.slot_channels()
        self.x = E.Act(self.seg, 2 * B, H, W, in_channels, name='input', cstride=slot if in_channels < slot else None)
        self.feats = E.build_encoder(self.seg, self.x, n=n)
        self.logits = E.build_decoder(self.seg, self.feats, n=n, num_classes=num_classes)
        gs = [0]
        for b in batch_sizes:
            gs.append(gs[-1] + b)
        self.rec = E.Plan(bank, dtype, B, gs, slope=slope)
        self.rec.pad_narrow = True
        self.rec.materialize_up = self.seg.materialize_up
        self.rec.materialize_pool = self.seg.materialize_pool
        self.rec.fused_bwd = self.seg.fused_bwd
        self.rec.fold_finalize = self.seg.fold_finalize
        self.rec.fold_wgrad_behind = self.seg.fold_wgrad_behind
        self.rec.fold_fwd_kinds = self.seg.fold_fwd_kinds
        self.rec.store_operands = self.seg.store_operands
        self.rec.store_min_c = self.seg.store_min_c
        lane = bool(budget and opt['rec_lane'])
        self.rec.side_cus = T.cu_budget(opt['rec_cus'], dev) if lane else 0    # its weight gradients run inline on its own lane
        self.rec.conv_cus = T.cu_budget(opt['rec_cus'], dev) if lane else 0
        self.rec.materialize_min_c = self.seg.materialize_min_c
        self.rec.materialize_dz_min_c = self.seg.materialize_dz_min_c
        self.rec.materialize_dz_wide = self.seg.materialize_dz_wide
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
        # TWO input slots: raw inputs (src / trg / lam), a RAM descriptor and a copy of the network input x each.  While step N trains
        # on slot k, the host uploads batch N+1 into slot 1-k (load_raw_next) and step N mixes it into x[1-k] on the restoration lane
        # while the encoder backward runs on the main lane (step(): the pipelined form below) -- the three latency-bound RAM launches
        # no longer open the next step alone.
        self.ram = None
        self._slot, self._x_ready, self._next_loaded = 0, False, False
        self.xbufs = [self.x.buf, torch.zeros_like(self.x.buf) if ram else self.x.buf]
        if ram:
            idt = torch.uint8 if ram == 'u8' else torch.float32
            self.raw_slots = [(torch.zeros(B, H, W, in_channels, dtype=idt, device=dev), torch.zeros(B, H, W, in_channels, dtype=idt, device=dev),
                               torch.ones(B, dtype=torch.float32, device=dev)) for _ in range(2)]
            self.rams = [R.RamMixer(B, H, W, dtype, dev, dataset) for _ in range(2)]
            self.rams[1].share_workspace(self.rams[0])               # never in flight together: a step mixes at most one batch
            for k in range(2):
                self.rams[k].bind(*self.raw_slots[k], self.xbufs[k][:B], self.xbufs[k][B:])
            self.ram = self.rams[0]
            self.src, self.trg, self.lam = self.raw_slots[0]         # the current slot's buffers (load_raw's destination)
        self.graph = None
        self._zero_args = None
        self._native = {}
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
        self.launch_threads = int(opt['launch_threads']) > 0 and self.fork      # (-1: DataParallelStep measures it; one process: one thread)
        self._slot1_keep = []
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
        a += self.seg.fwd[:fdec]                                     # encoder
        a.append(E.sync_op('fork', 'rec'))
        # The restoration lane's STAND-ALONE weight gradients (its >= 64-channel and 1x1 layers, their split reductions, the bias
        # column sum) are taken out of its dgrad chain and enqueued behind the join (opt rec_wgrad_late): nothing but Adam needs them,
        # while the main lane waits at the join for the chain (scripts/join_probe.py: main arrives at 3.07 ms, the lane at 3.16).
        # They then run on the same stream during the encoder backward, when that lane has nothing else to do.
        late = bool(self.opt['rec_wgrad_late'])
        is_late = lambda op: late and len(op) > 2 and bool(op[2].get('side'))
        rec_bwd = self.rec.bwd[:-1]
        rec_branch = E.tag_lane(self.rec.fwd + [rec_loss] + [op for op in rec_bwd if not is_late(op)], 'rec')
        rec_late = E.tag_lane([op for op in rec_bwd if is_late(op)], 'rec')
        dec_branch = self.seg.fwd[fdec:] + [seg_loss] + self.seg.bwd[:split]
        a += E.interleave(dec_branch, rec_branch)
        a.append(E.sync_op('join', 'rec'))
        a.append(self.rec.bwd[-1])                                   # rec.convu4.conv1 dgrad: += into the bottleneck gradient
        assert self.rec.bwd[-1][2].get('what') == 'dgrad', self.rec.bwd[-1][2]
        if rec_late:
            a.append(E.sync_op('fork', 'rec'))                       # (re-opens the lane: the join in front of Adam covers it)
            a += rec_late
        b = list(self.seg.bwd[split:])
        # encoder backward in two parts (data parallel: the deep levels hold 98 % of the encoder's parameters and finish
        # first, so their gradients are all-reduced while the 200x200 / 400x400 levels are still running)
        cut = self.seg.bwd_node_start.get(('enc', 'convd2.conv3'), split) - split
        self.seg_b1, self.seg_b2 = b[:cut], b[cut:]
        self.enc_deep_offset = min(off for (m, k), (off, _) in self.bank.index.items() if m == 'enc' and k.startswith('convd3.'))
        c = [(lib.rd_adam_step, (C.byref(self.ad),)), self.wpack.refresh_op()]
        # Segment lists per input slot k.  Three launches read the network input: the first conv's forward, its weight gradient, and the
        # restoration loss (target = the un-mixed images); slot 1's lists carry copies of those descriptors that point at x[1].
        #   seg_ram_s<k>      RAM of slot k -> x[k] as a head segment (the classical order: mix, then train on it)
        #   seg_a_noram_s<k>  forward, losses, decoder backwards on x[k]
        #   seg_b_s<k>        encoder backward (seg_b1_s<k> + seg_b2_s<k>: the data-parallel split)
        #   seg_b_pf_s<k> / seg_b1_pf_s<k>   the same with the NEXT batch's RAM (slot 1-k -> x[1-k]) forked onto the restoration lane in
        #                   front of it: that lane has finished its branch by then and is idle for the whole encoder backward; the
        #                   final join in front of Adam covers it
        for k in range(2):
            ak, bk = (a, b) if k == 0 else ([self._on_slot1(op) for op in a], [self._on_slot1(op) for op in b])
            setattr(self, 'seg_a_noram_s%d' % k, ak)
            setattr(self, 'seg_b_s%d' % k, bk)
            setattr(self, 'seg_b1_s%d' % k, bk[:cut])
            setattr(self, 'seg_b2_s%d' % k, bk[cut:])
            setattr(self, 'seg_ram_s%d' % k, [self.rams[k].op()] if self.ram is not None else [])
        for k in range(2):
            pf = ([E.sync_op('fork', 'rec')] + E.tag_lane([self.rams[1 - k].op()], 'rec')) if self.ram is not None else []
            setattr(self, 'seg_b_pf_s%d' % k, pf + getattr(self, 'seg_b_s%d' % k))
            setattr(self, 'seg_b1_pf_s%d' % k, pf + getattr(self, 'seg_b1_s%d' % k))
        # the classical segments on slot 0 (RAM at the head of A): hipGraph capture, the scripts, bench.py's instrumented passes
        self.seg_a_noram = a
        self.seg_a, self.seg_b, self.seg_c = self.seg_ram_s0 + a, b, c
        return self.seg_a + b + c

    def _on_slot1(self, op):
        """The launch `op` with every reference to the network input x[0] replaced by x[1] (a copy of its descriptor); other launches
        are returned as they are."""
        x0, x1 = self.xbufs[0].data_ptr(), self.xbufs[1].data_ptr()
        if op[0] is None or x0 == x1:
            return op
        args, hit = [], False
        for v in op[1]:
            if hasattr(v, '_obj') and isinstance(v._obj, (L.RdConv, L.RdWgrad)):
                d = type(v._obj)()
                C.memmove(C.byref(d), C.byref(v._obj), C.sizeof(d))
                srcs = list(d.src) if isinstance(d, L.RdConv) else list(d.a)
                found = False
                for sdesc in srcs:
                    if sdesc.ptr == x0:
                        sdesc.ptr = x1
                        found = True
                if found:
                    self._slot1_keep.append(d)
                    args.append(C.byref(d))
                    hit = True
                    continue
            elif isinstance(v, int) and v == x0:
                args.append(x1)
                hit = True
                continue
            args.append(v)
        return (op[0], tuple(args)) + tuple(op[2:]) if hit else op

    def lane_layout(self):
        """Which stream carries which lane and whether it was measured to run beside the others (streams.pick_lanes): diagnostic
        for the bench line -- on a new stack (the first 8-GPU run) a lane that shares a hardware queue shows up here, not as an
        unexplained slow step."""
        out = {'main': {'stream': int(torch.cuda.current_stream(self.bank.device).cuda_stream)}}
        for name, st in self.lanes().items():
            out[name] = {'stream': int(st.cuda_stream), 'verified_concurrent': bool(streams.verified.get(id(st), False))}
        out['budgets'] = {'side_cus': int(self.seg.side_cus), 'rec_cus': int(self.rec.conv_cus), 'dgrad_cus': int(self.seg.dgrad_cus)}
        return out

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
        xb = self.x_current()
        half = xb[B:]
        L.check(lib.rd_nchw_to_nhwc(img_nchw.data_ptr(), xb.data_ptr(), B, self.c, self.H, self.W, self.x.Cs, self.dt, s), 'load img')
        L.check(lib.rd_nchw_to_nhwc(img_freq_nchw.data_ptr(), half.data_ptr(), B, self.c, self.H, self.W, self.x.Cs, self.dt, s), 'load img_freq')

    def load_raw(self, src_nhwc, trg_nhwc, lam):
        """RAM inputs: what Fundus_Multi.__getitem__ holds before the FFTs (fundus.py:209-212): the
        transformed image and the partner image as HWC arrays, and the mix ratio.  The buffers are uint8 when the step
        was built with ram='u8' (decoded PNG pixels) and float32 otherwise; a float image handed to a uint8 step would be
        truncated / wrapped by the copy where the reference mixes it as float32, so that combination is refused."""
        self._copy_raw(self._slot, src_nhwc, trg_nhwc, lam)
        self._x_ready, self._next_loaded = False, False

    def load_raw_next(self, src_nhwc, trg_nhwc, lam):
        """The RAM inputs of the NEXT step, into the other input slot: the following step() then mixes them on the restoration lane while
        its encoder backward runs, instead of opening the step after it with three latency-bound RAM launches.  Call order per
        iteration: load_raw_next(batch N+1); step(); load_target(mask N+1) -- the target buffer is single: it is read by step N's
        loss, so the next mask is uploaded after step N has been enqueued (stream order does the rest)."""
        self._copy_raw(1 - self._slot, src_nhwc, trg_nhwc, lam)
        self._next_loaded = True

    def reuse_next(self):
        """The other input slot already holds the next batch (inputs resident in HBM: bench.py steps over one synthetic batch that
        was loaded into both slots): the following step() runs pipelined without a copy."""
        self._next_loaded = True

    def _copy_raw(self, slot, src_nhwc, trg_nhwc, lam):
        src, trg, lm = self.raw_slots[slot]
        for name, t in (('src', src_nhwc), ('trg', trg_nhwc)):
            if src.dtype == torch.uint8 and t.dtype != torch.uint8:
                raise TypeError('%s images are %s but this TrainStep was built with ram=\'u8\' (uint8 buffers): pass uint8 '
                                'pixels or build the step with ram=True (float32 buffers)' % (name, t.dtype))
        # asynchronous only for sources that are already on the device: a pinned HOST buffer the caller reuses after this call
        # (DataLoader pinned memory, a bench loop) would race with a non-blocking H2D copy
        for dst, t in ((src, src_nhwc), (trg, trg_nhwc), (lm, lam)):
            dst.copy_(t, non_blocking=bool(t.is_cuda))

    def load_target(self, mask):
        self.target.copy_(mask)

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    # ---- execution
    def zero(self, stream=None):
        """optimizer.zero_grad() (train.py:285) + the BatchNorm sum arenas, through the library (rd_zero)."""
        fn, args = self.zero_op()
        L.check(fn(*args, self._stream() if stream is None else stream), 'rd_zero')

    def run_segment(self, ops, main=None, lanes=None, wrap=None):
        """One segment over the lanes, entry by entry from Python (Plan.run_lanes: the instrumented path -- `wrap` may time each
        launch); every lane it used is joined before returning.  Plain execution goes through launch() below."""
        main = torch.cuda.current_stream() if main is None else main
        lanes = self.lanes() if lanes is None else lanes
        for lane in E.Plan.run_lanes(ops, main, lanes, wrap):
            main.wait_stream(lanes[lane])

    def zero_op(self):
        if self._zero_args is None:
            ts = (self.seg.stat_arena, self.rec.stat_arena, self.bank.grads)
            self._zero_args = ((L.vp * 3)(*[t.data_ptr() for t in ts]), (L.i64 * 3)(*[t.numel() * t.element_size() for t in ts]))
        return (L.lib().rd_zero, (self._zero_args[0], self._zero_args[1], 3))

    def native_list(self, names, lanes, join_before_last=False):
        """The segments `names` (attribute names: 'seg_a', 'seg_b', ...; 'zero' = the per-step reset) as ONE native launch list
        (engine.LaunchList -> rd_run_list), compiled once per lane configuration and rebuilt when a segment list has been replaced
        (scripts/ablate_step.py, bench.py's ablation) or changed length.  join_before_last: every lane is joined in front of the
        last segment (the optimizer must see all gradients)."""
        lane_names = tuple(lanes.keys())
        key = (tuple(names), lane_names, join_before_last)
        lists = [[self.zero_op()] if nm == 'zero' else getattr(self, nm) for nm in names]
        ent = self._native.get(key)
        # stale when a segment attribute now holds another list object, or the same object with another length
        fresh = ent is not None and all(nm == 'zero' or (src is l and n == len(l)) for nm, l, (src, n) in zip(names, lists, ent[1]))
        if not fresh:
            ops = []
            for i, l in enumerate(lists):
                if join_before_last and i == len(lists) - 1:
                    ops += [E.sync_op('join', ln) for ln in lane_names]
                ops += l
            ent = (E.LaunchList(ops, lane_names), [(l, len(l)) for l in lists])
            self._native[key] = ent
        return ent[0]

    def launch(self, names, main=None, lanes=None, open_mask=0, join=True, join_before_last=False):
        """Enqueue segments through the native launch loop: ONE ctypes call.  Returns (list, bitmask of the lanes still open)."""
        main = torch.cuda.current_stream() if main is None else main
        lanes = self.lanes() if lanes is None else lanes
        ll = self.native_list(names, lanes, join_before_last)
        L.lib().rd_run_list_threads(1 if self.launch_threads else 0)      # process-wide switch: set per call, this step's choice
        mask = ll.run(main, lanes, open_mask)
        if join and mask:
            ll.join(main, lanes, mask)
            mask = 0
        return ll, mask

    def run_eager(self, lanes=None):
        """The classical step on input slot 0: RAM, forward, backward, ONE join of the weight-gradient lane right before Adam, Adam +
        repack -- one native call (rd_run_list)."""
        self.launch(('zero', 'seg_a', 'seg_b', 'seg_c'), lanes=lanes, join_before_last=True)

    def head_names(self):
        """Segment names of the step's first part: the reset, RAM of the current slot unless the previous step has mixed it already,
        forward + losses + decoder backwards on the current slot's x."""
        k = self._slot
        return ('zero',) + (() if (self._x_ready or self.ram is None) else ('seg_ram_s%d' % k,)) + ('seg_a_noram_s%d' % k,)

    def backward_names(self, split=False):
        """Encoder backward on the current slot -- with the next batch's RAM beside it when load_raw_next() has provided one;
        split: the data-parallel step's two parts."""
        k = self._slot
        pf = '_pf' if (self._next_loaded and self.ram is not None) else ''
        return ('seg_b1%s_s%d' % (pf, k), 'seg_b2_s%d' % k) if split else ('seg_b%s_s%d' % (pf, k),)

    def x_current(self):
        """The network input [img ; img_freq] of the current slot (NHWC, channel vector padded to one 16-byte slot)."""
        return self.xbufs[self._slot]

    def advance(self):
        """Bookkeeping after a step has been enqueued: a pipelined step has left the next batch's x ready and flips the slots."""
        if self._next_loaded and self.ram is not None:
            self._slot, self._x_ready = 1 - self._slot, True
            self.src, self.trg, self.lam = self.raw_slots[self._slot]
        else:
            self._x_ready = False
        self._next_loaded = False

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
        """One training step on the current input slot.  If load_raw_next() has been called, the step also mixes the next batch
        (RAM on the restoration lane, beside the encoder backward); a captured graph always replays the classical list on slot 0."""
        if self.graph is not None:
            if self._slot != 0 or self._x_ready or self._next_loaded:
                raise RuntimeError('a captured hipGraph replays the classical step on input slot 0: do not mix it with load_raw_next()')
            self.graph.replay()
            return
        self.launch(self.head_names() + self.backward_names() + ('seg_c',), join_before_last=True)
        self.advance()

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
