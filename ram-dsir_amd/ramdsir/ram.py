"""Random Amplitude Mixup on the GPU (host side).  Mirrors the reference's per-sample numpy trio
(code/dataset/fundus.py:13-61) at batch level: the DataLoader-side code only picks the partner image and
the ratio lam = random.randint(1,10)/10 (fundus.py:35,201-209); the FFTs run in rd_ram_mix."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L
from . import engine as E


def window_half_width(h, w, ratio=0.1):
    """b = floor(min(h,w)*L), fundus.py:26."""
    return int(math.floor(min(h, w) * ratio))


def _twiddle(n, device):
    k = np.arange(n, dtype=np.float64)
    t = np.stack([np.cos(2 * np.pi * k / n), -np.sin(2 * np.pi * k / n)], -1).astype(np.float32)
    return torch.from_numpy(t).to(device)


class RamMixer:
    """Plan for one (B, H, W) geometry.  `fundus`: inputs on the 0..255 scale, clip [0,255], /127.5-1
    (fundus.py:215-225); `prostate`: inputs in [-1,1], clip [-1,1] (prostate.py:188)."""

    def __init__(self, B, H, W, dtype, device, dataset='fundus', ratio=0.1):
        self.B, self.H, self.W = B, H, W
        self.dt = L.RD_BF16 if dtype == torch.bfloat16 else L.RD_F32
        self.dtype = dtype
        self.b = window_half_width(H, W, ratio)
        lib = L.lib()
        self.ws = E.workspace(lib.rd_ram_workspace(B, H, W, self.b) // 4, device)
        self.tw_w, self.tw_h = _twiddle(W, device), _twiddle(H, device)
        p = L.RdRam()
        p.workspace, p.tw_w, p.tw_h = self.ws.data_ptr(), self.tw_w.data_ptr(), self.tw_h.data_ptr()
        p.B, p.H, p.W, p.C, p.b = B, H, W, 3, self.b
        if dataset == 'fundus':
            p.clip_lo, p.clip_hi, p.scale, p.offset = 0.0, 255.0, 1.0 / 127.5, -1.0
        else:
            p.clip_lo, p.clip_hi, p.scale, p.offset = -1.0, 1.0, 1.0, 0.0
        self.p = p

    def bind(self, src, trg, lam, out_img, out_freq):
        """src/trg: fp32 NHWC [B,H,W,3]; lam: fp32 [B]; outputs: NHWC `dtype` [B,H,W,Cs] with Cs >= 3 (views are
        fine; channels 3..Cs-1 are never written)."""
        self.p.src, self.p.trg, self.p.lam = src.data_ptr(), trg.data_ptr(), lam.data_ptr()
        self.p.out_img, self.p.out_freq = out_img.data_ptr(), out_freq.data_ptr()
        assert out_img.shape[-1] == out_freq.shape[-1] >= 3
        self.p.out_cstride = out_img.shape[-1]
        self._keep = (src, trg, lam, out_img, out_freq)

    def op(self):
        return (L.lib().rd_ram_mix, (C.byref(self.p), self.dt))

    def run(self, stream=None):
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        fn, args = self.op()
        L.check(fn(*args, stream), 'rd_ram_mix')


def source_to_target_freq_batch(src_nhwc, trg_nhwc, lam, dataset='fundus', dtype=torch.float32):
    """Convenience wrapper: returns (img, img_freq) as NCHW fp32 tensors in [-1,1], like the tuple the
    reference's Fundus_Multi.__getitem__ yields after collation (fundus.py:240)."""
    B, H, W, _ = src_nhwc.shape
    dev = src_nhwc.device
    m = RamMixer(B, H, W, dtype, dev, dataset)
    oi = torch.empty(B, H, W, 3, dtype=dtype, device=dev)
    of = torch.empty(B, H, W, 3, dtype=dtype, device=dev)
    m.bind(src_nhwc.float().contiguous(), trg_nhwc.float().contiguous(), lam.float().contiguous(), oi, of)
    m.run()
    return oi.float().permute(0, 3, 1, 2).contiguous(), of.float().permute(0, 3, 1, 2).contiguous()
