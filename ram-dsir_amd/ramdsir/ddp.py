"""Data-parallel training: one process per GPU, rank-local BatchNorm statistics (the reference's
nn.DataParallel keeps per-replica BN too, code/train.py:205-208), ONE exchange per step: the flat fp32
gradient arena (3.8 M floats = 15.2 MB) is averaged with RCCL all-reduce in three buckets that follow the order
in which the backward produces them --
  bucket 0 = seg-decoder + rec-decoder gradients, ready when segment A ends: reduced on the communication stream
             while the encoder backward runs;
  bucket 1 = encoder levels 3-5 (98 % of the encoder's parameters), ready after the first part of the encoder
             backward (segment B1, the 100x100 and smaller levels): reduced while the 200x200 / 400x400 levels
             (segment B2, most of the encoder's backward TIME) are still running;
  bucket 2 = encoder levels 1-2 (28 K floats): the only exchange that is not hidden, and it is latency-sized;
all are waited for before Adam (segment C).  On a fully connected xGMI node the payload is latency-bound
(SURVEY.md section 5), hence a few large buckets rather than per-tensor all-reduces.
"""
import torch
import torch.distributed as dist

from . import engine as E


class GradBuckets:
    """Bucketing + averaging of a flat gradient tensor; pure torch.distributed (runs on gloo/CPU in tests).
    boundaries: ascending offsets [0, b1, ..., n]; bucket i = flat[boundaries[i]:boundaries[i+1]]."""

    def __init__(self, flat_grads, boundaries, group=None):
        self.flat = flat_grads
        self.bounds = list(boundaries)
        assert self.bounds[0] == 0 and all(a <= b for a, b in zip(self.bounds, self.bounds[1:])) and self.bounds[-1] <= flat_grads.numel()
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.views = [self.flat[a:b] for a, b in zip(self.bounds, self.bounds[1:])]
        self._avg = dist.ReduceOp.AVG if (dist.is_initialized() and dist.get_backend(group) == 'nccl') else dist.ReduceOp.SUM

    def __len__(self):
        return len(self.views)

    def reduce(self, i, async_op=True):
        """Average bucket i over the ranks; returns the work handle (None when there is nothing to wait for)."""
        t = self.views[i]
        if t.numel() == 0 or (self.world == 1 and not dist.is_initialized()):
            return None
        w = dist.all_reduce(t, op=self._avg, group=self.group, async_op=async_op)
        if self._avg == dist.ReduceOp.SUM:
            if async_op:
                w.wait()
                w = None
            t.div_(self.world)
        return w


class DataParallelStep:
    """Wraps a TrainStep: four hipGraphs (segments A, B1, B2, C) with the three all-reduces between them."""

    def __init__(self, ts, group=None):
        self.ts = ts
        b = ts.bank
        n_enc = b.module_range['enc'][1]
        # arena order = parameters() order: encoder levels 1..5, then the decoders
        self.buckets = GradBuckets(b.grads, [0, ts.enc_deep_offset, n_enc, b.n], group)
        # The collectives are LAUNCHED from the weight-gradient lane (RCCL runs them on its own stream and only makes that
        # stream wait for the launching one): a separate communication stream would be the fifth stream of the step, and
        # HIP gives a process four hardware queues -- a fifth one aliases a lane (streams.py; 5.9 -> 9-11 ms/step measured)
        choice = int(ts.opt['ddp_own_comm_stream'])
        self._own = None
        self.comm_choice = {'own_comm_stream': bool(choice > 0), 'how': 'option'}
        self.comm = self._comm_stream(choice > 0)
        self.graphs = None
        if choice < 0:
            self.pick_comm()
        self.launch_choice = {'launch_threads': bool(ts.launch_threads), 'how': 'option'}
        if int(ts.opt['launch_threads']) < 0:
            self.pick_launch_threads()

    def _comm_stream(self, own):
        if own:
            if self._own is None:
                self._own = torch.cuda.Stream(device=self.ts.bank.device)
            return self._own
        return self.ts.side[0]

    def pick_comm(self, steps=4):
        """Launch the bucket exchanges from the weight-gradient lane or from a stream of their own?  Decided the way streams.pick_lanes
        decides the lanes: by measurement, here on the real process group -- `steps` eager steps each way (after one warm-up each) on
        whatever the input buffers hold, state restored afterwards; every rank takes the choice that is faster for the SLOWEST rank
        (one small all-reduce).  With one rank there is nothing to exchange with: the lane."""
        import time
        ts = self.ts
        if self.buckets.world <= 1 or not dist.is_initialized():
            self.comm_choice = {'own_comm_stream': False, 'how': 'single rank: the weight-gradient lane'}
            self.comm = self._comm_stream(False)
            return
        saved, state = ts._snapshot(), (ts._slot, ts._x_ready, ts._next_loaded)
        times = []
        for own in (False, True):
            self.comm = self._comm_stream(own)
            ts._restore(saved)
            self.step()
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) / steps * 1e3)
        t = torch.tensor(times, dtype=torch.float64, device=ts.bank.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        times = [float(v) for v in t]
        own = times[1] < times[0]
        self.comm = self._comm_stream(own)
        self.comm_choice = {'own_comm_stream': bool(own), 'how': 'measured', 'ms_per_step_lane': round(times[0], 3), 'ms_per_step_own': round(times[1], 3)}
        ts._restore(saved)
        ts._slot, ts._x_ready, ts._next_loaded = state
        ts.src, ts.trg, ts.lam = ts.raw_slots[ts._slot] if ts.ram is not None else (None, None, None)
        torch.cuda.synchronize()

    def pick_launch_threads(self, steps=6):
        """One host thread or one per lane for the enqueue (rd_run_list_threads)?  Eight ranks share one host, and a rank's enqueue is more
        than half of its step time, so the answer depends on how many cores the ranks leave each other: measured on the real process group,
        like the communication stream above -- `steps` eager steps each way, the slowest rank decides, state restored."""
        import time
        ts = self.ts
        if not ts.fork:
            self.launch_choice = {'launch_threads': False, 'how': 'one stream: nothing to parallelise'}
            return
        saved, state = ts._snapshot(), (ts._slot, ts._x_ready, ts._next_loaded)
        times = []
        for threads in (False, True):
            ts.launch_threads = threads
            ts._restore(saved)
            self.step()
            torch.cuda.synchronize()
            if dist.is_initialized():
                dist.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - t0) / steps * 1e3)
        t = torch.tensor(times, dtype=torch.float64, device=ts.bank.device)
        if dist.is_initialized() and self.buckets.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        times = [float(v) for v in t]
        ts.launch_threads = bool(times[1] < 0.98 * times[0])          # worker threads only for a measurable gain
        self.launch_choice = {'launch_threads': ts.launch_threads, 'how': 'measured', 'ms_per_step_one_thread': round(times[0], 3),
                              'ms_per_step_lane_threads': round(times[1], 3)}
        ts._restore(saved)
        ts._slot, ts._x_ready, ts._next_loaded = state
        ts.src, ts.trg, ts.lam = ts.raw_slots[ts._slot] if ts.ram is not None else (None, None, None)
        torch.cuda.synchronize()

    def _segments(self):
        ts = self.ts
        return (ts.seg_a, ts.seg_b1, ts.seg_b2, ts.seg_c)

    def capture(self):
        ts = self.ts
        ts._refuse_budgets_on_one_chain('DataParallelStep.capture()')
        ts.wpack.refresh(ts._stream())
        torch.cuda.synchronize()
        saved = ts._snapshot()
        ts.run_eager()                              # warm-up (lazy kernel loading) outside capture
        torch.cuda.synchronize()
        ts._restore(saved)
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        self.graphs = []
        with torch.cuda.stream(st):
            for i, seg in enumerate(self._segments()):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    if i == 0:
                        ts.zero()
                    ts.run_segment(seg, st, lanes={})      # one chain per graph: forked branches replay slower
                self.graphs.append(g)
        torch.cuda.synchronize()
        ts._restore(saved)
        torch.cuda.synchronize()

    def _names(self, i):
        """Segment i of the eager step by name (TrainStep.launch): A = reset + [RAM] + forward + decoder backwards, B1 / B2 = encoder
        backward, C = Adam + repack -- with the next batch's RAM beside it when load_raw_next() has provided one (step.py)."""
        ts = self.ts
        b1, b2 = ts.backward_names(split=True)
        return (ts.head_names(), (b1,), (b2,), ('seg_c',))[i]

    def _run(self, i, main):
        if self.graphs is not None:
            self.graphs[i].replay()
        else:
            self.ts.launch(self._names(i), main)

    def step(self):
        main = torch.cuda.current_stream()
        works = []
        if self.graphs is not None and (self.ts._slot != 0 or self.ts._x_ready or self.ts._next_loaded):
            raise RuntimeError('captured graphs replay the classical step on input slot 0: do not mix them with load_raw_next()')
        # segment -> bucket that is complete when it ends: A -> decoders (2), B1 -> deep encoder (1), B2 -> shallow (0)
        plan = ((0, 2), (1, 1), (2, 0))
        if self.graphs is not None:
            for seg_i, bucket in plan:
                self._run(seg_i, main)
                self.comm.wait_stream(main)
                with torch.cuda.stream(self.comm):
                    works.append(self.buckets.reduce(bucket, async_op=True))
        else:
            # eager: the weight-gradient stream is NOT joined into the main stream at the segment boundaries (that
            # would serialise the dgrad chain behind the weight gradients three times per step); a bucket's exchange
            # waits for the main stream and for every lane with work outstanding -- the lanes are in-order queues, so
            # their state at this point covers exactly the gradients of the segments launched so far
            # the segments go through the same native launch loop as the single-GPU step (TrainStep.launch -> rd_run_list), the open
            # lanes carried from one segment to the next as its bitmask
            ts = self.ts
            lanes, mask, ll = ts.lanes(), 0, None
            for seg_i, bucket in plan:
                ll, mask = ts.launch(self._names(seg_i), main, lanes, open_mask=mask, join=False)
                self.comm.wait_stream(main)
                for k, name in enumerate(ll.lane_names):
                    if mask & (1 << (k + 1)) and lanes[name] is not self.comm:
                        self.comm.wait_stream(lanes[name])
                with torch.cuda.stream(self.comm):
                    works.append(self.buckets.reduce(bucket, async_op=True))
            if ll is not None:
                ll.join(main, lanes, mask)
        for w in works:
            if w is not None:
                w.wait()
        main.wait_stream(self.comm)
        self._run(3, main)
        if self.graphs is None:
            self.ts.advance()
