"""Data-parallel training: one process per GPU, rank-local BatchNorm statistics (the reference's
nn.DataParallel keeps per-replica BN too, code/train.py:205-208), ONE exchange per step: the flat fp32
gradient arena (3.8 M floats = 15.2 MB) is averaged with RCCL all-reduce in two buckets --
  bucket 0 = seg-decoder + rec-decoder gradients, ready when segment A ends, reduced on a side stream
             while the encoder backward (segment B) runs;
  bucket 1 = encoder gradients, reduced after segment B;
both are waited for before Adam (segment C).  On a fully connected xGMI node the payload is latency-bound
(SURVEY.md section 5), hence two large buckets rather than per-tensor all-reduces.
"""
import torch
import torch.distributed as dist

from . import engine as E


class GradBuckets:
    """Bucketing + averaging of a flat gradient tensor; pure torch.distributed (runs on gloo/CPU in tests)."""

    def __init__(self, flat_grads, boundaries, group=None):
        # boundaries: [0, n_enc, n_total] -> bucket 1 = [0, n_enc) (encoder), bucket 0 = [n_enc, n_total)
        self.flat = flat_grads
        self.n_enc, self.n = boundaries[1], boundaries[2]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.late = self.flat[self.n_enc:self.n]      # decoder + rec decoder
        self.early = self.flat[0:self.n_enc]          # encoder
        self._avg = dist.ReduceOp.AVG if (dist.is_initialized() and dist.get_backend(group) == 'nccl') else dist.ReduceOp.SUM

    def _reduce(self, t, async_op):
        if self.world == 1 and not dist.is_initialized():
            return None
        w = dist.all_reduce(t, op=self._avg, group=self.group, async_op=async_op)
        if self._avg == dist.ReduceOp.SUM:
            if async_op:
                w.wait()
                w = None
            t.div_(self.world)
        return w

    def reduce_decoder_side(self, async_op=True):
        return self._reduce(self.late, async_op)

    def reduce_encoder(self, async_op=True):
        return self._reduce(self.early, async_op)


class DataParallelStep:
    """Wraps a TrainStep: three hipGraphs (segments A, B, C) with the two all-reduces between them."""

    def __init__(self, ts, group=None):
        self.ts = ts
        b = ts.bank
        self.buckets = GradBuckets(b.grads, [0, b.module_range['enc'][1], b.n], group)
        self.comm = torch.cuda.Stream()
        self.graphs = None

    def capture(self):
        ts = self.ts
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
            for i, seg in enumerate((ts.seg_a, ts.seg_b, ts.seg_c)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    if i == 0:
                        ts.zero()
                    ts.run_segment(seg, st)
                self.graphs.append(g)
        torch.cuda.synchronize()
        ts._restore(saved)
        torch.cuda.synchronize()

    def step(self):
        ts = self.ts
        main = torch.cuda.current_stream()
        if self.graphs is not None:
            self.graphs[0].replay()
        else:
            ts.zero()
            ts.run_segment(ts.seg_a, main)
        self.comm.wait_stream(main)
        with torch.cuda.stream(self.comm):
            w0 = self.buckets.reduce_decoder_side(async_op=True)
        if self.graphs is not None:
            self.graphs[1].replay()
        else:
            ts.run_segment(ts.seg_b, main)
        w1 = self.buckets.reduce_encoder(async_op=True)
        for w in (w0, w1):
            if w is not None:
                w.wait()
        main.wait_stream(self.comm)
        if self.graphs is not None:
            self.graphs[2].replay()
        else:
            ts.run_segment(ts.seg_c, main)
