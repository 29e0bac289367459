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


_TABLES = {}


def dft_tables(H, W, b, device):
    """Coefficient tables of the matrix-core path for one geometry (rd_ram_dft_tables, built once per process and device), or None
    when the geometry runs the FFT kernels."""
    key = (H, W, b, str(device))
    if key not in _TABLES:
        lib = L.lib()
        n = int(lib.rd_ram_dft_tables_bytes(H, W, b))
        t = None
        if n > 0:
            t = torch.empty(n, dtype=torch.uint8, device=device)
            L.check(lib.rd_ram_dft_tables(t.data_ptr(), H, W, b, torch.cuda.current_stream(device).cuda_stream), 'rd_ram_dft_tables')
            torch.cuda.current_stream(device).synchronize()
        _TABLES[key] = t
    return _TABLES[key]


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
        self.tables = dft_tables(H, W, self.b, device)
        p = L.RdRam()
        p.workspace, p.tw_w, p.tw_h = self.ws.data_ptr(), self.tw_w.data_ptr(), self.tw_h.data_ptr()
        p.dft_tables = self.tables.data_ptr() if self.tables is not None else None
        p.B, p.H, p.W, p.C, p.b = B, H, W, 3, self.b
        if dataset == 'fundus':
            # x / 127.5 - 1 as a DIVISION: bit for bit numpy's `img /= 127.5; img -= 1.0` (fundus.py:217-218)
            p.clip_lo, p.clip_hi, p.scale, p.div, p.offset = 0.0, 255.0, 1.0 / 127.5, 127.5, -1.0
        else:
            p.clip_lo, p.clip_hi, p.scale, p.div, p.offset = -1.0, 1.0, 1.0, 0.0, 0.0
        self.p = p

    def share_workspace(self, other):
        """Use another mixer's workspace and twiddle tables (same geometry; the two are never in flight together)."""
        assert (self.B, self.H, self.W, self.b) == (other.B, other.H, other.W, other.b)
        self.ws, self.tw_w, self.tw_h, self.tables = other.ws, other.tw_w, other.tw_h, other.tables
        self.p.dft_tables = self.tables.data_ptr() if self.tables is not None else None
        self.p.workspace, self.p.tw_w, self.p.tw_h = self.ws.data_ptr(), self.tw_w.data_ptr(), self.tw_h.data_ptr()

    def bind(self, src, trg, lam, out_img, out_freq, trg_amp=None):
        """src/trg: NHWC [B,H,W,3], both fp32 or both uint8 (decoded PNG pixels); lam: fp32 [B]; outputs: NHWC `dtype`
        [B,H,W,Cs] with Cs == 3 (dense) or one 16-byte slot (channels 3.. are then written as zeros).  trg_amp: fp32
        [B,3,H,W] amplitude spectra of the partners instead of their images (trg may then be None)."""
        assert src.is_contiguous() and lam.dtype == torch.float32 and src.dtype in (torch.float32, torch.uint8)
        assert trg is None or (trg.is_contiguous() and trg.dtype == src.dtype)
        self.p.src, self.p.lam = src.data_ptr(), lam.data_ptr()
        self.p.trg = trg.data_ptr() if trg is not None else None
        self.p.trg_amp = trg_amp.data_ptr() if trg_amp is not None else None
        self.p.src_u8 = 1 if src.dtype == torch.uint8 else 0
        self.p.out_img, self.p.out_freq = out_img.data_ptr(), out_freq.data_ptr()
        assert out_img.shape[-1] == out_freq.shape[-1] >= 3
        self.p.out_cstride = out_img.shape[-1]
        self._keep = (src, trg, lam, out_img, out_freq, trg_amp)

    def op(self):
        # meta for bench.py's family table: algorithmic bytes by SURVEY.md 8(d)'s convention = source + partner + output once as
        # fp32, 12 * C * H * W per image (the uint8 pipeline reads less and the bf16 output writes less; the convention is kept so
        # that the figure is comparable with the reference's numpy path, code/dataset/fundus.py:13-61); ~62 MFLOP per 400 x 400 image
        # bytes_as_built: what the three kernels have to move -- source + partner pixels once for the row pass (1 or 4 bytes per value) and
        # the source again for the output pass, the kept row bins [2B][3][H][KP] complex64 written and read, the column results
        # [B][3][H][KP] written and read, and the two output tensors at their pixel stride
        px = self.H * self.W * self.B
        ebytes = 1 if self.p.src_u8 else 4
        kp = (self.b + 1 + 3) // 4 * 4
        spec = 3 * self.H * kp * 8 * self.B
        out_b = 2 * px * max(int(self.p.out_cstride), 3) * (2 if self.dtype == torch.bfloat16 else 4)
        return (L.lib().rd_ram_mix, (C.byref(self.p), self.dt),
                dict(kernel='ram', what='mix', layer='ram', bytes=12 * 3 * self.H * self.W * self.B,
                     bytes_as_built=3 * px * 3 * ebytes + 2 * (2 * spec) + 2 * spec + out_b,
                     flops=int(62e6 * self.H * self.W / 160000.0) * self.B))

    def run(self, stream=None):
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        fn, args = self.op()[:2]
        L.check(fn(*args, stream), 'rd_ram_mix')


def source_to_target_freq_batch(src_nhwc, trg_nhwc, lam, dataset='fundus', dtype=torch.float32):
    """Convenience wrapper: returns (img, img_freq) as NCHW fp32 tensors in [-1,1], like the tuple the
    reference's Fundus_Multi.__getitem__ yields after collation (fundus.py:240)."""
    B, H, W, _ = src_nhwc.shape
    dev = src_nhwc.device
    m = RamMixer(B, H, W, dtype, dev, dataset)
    oi = torch.empty(B, H, W, 3, dtype=dtype, device=dev)
    of = torch.empty(B, H, W, 3, dtype=dtype, device=dev)
    if not (src_nhwc.dtype == torch.uint8 and trg_nhwc.dtype == torch.uint8):
        src_nhwc, trg_nhwc = src_nhwc.float(), trg_nhwc.float()
    m.bind(src_nhwc.contiguous(), trg_nhwc.contiguous(), lam.float().contiguous(), oi, of)
    m.run()
    return oi.float().permute(0, 3, 1, 2).contiguous(), of.float().permute(0, 3, 1, 2).contiguous()


# ---------------------------------------------------------------------------------------------------------------
# The reference's three free functions on the GPU, arrays in / arrays out (code/dataset/fundus.py:13-61).  The training
# path never calls these (it runs rd_ram_mix per batch); they keep code written against the reference API working.
def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError('the RAM kernels need a GPU: there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def _stream():
    return torch.cuda.current_stream().cuda_stream


def extract_amp_spectrum_gpu(img_chw):
    """|fft2(img)| over the last two axes (fundus.py:13-19); float32 [C,H,W] device tensor."""
    dev = _dev()
    x = torch.as_tensor(img_chw, dtype=torch.float32).to(dev).contiguous()
    Cc, H, W = x.shape
    lib = L.lib()
    ws = E.workspace(lib.rd_ram_amp_workspace(Cc, H, W) // 4, dev)
    tw_w, tw_h = _twiddle(W, dev), _twiddle(H, dev)
    out = torch.empty(Cc, H, W, dtype=torch.float32, device=dev)
    L.check(lib.rd_ram_amp(x.data_ptr(), out.data_ptr(), Cc, H, W, ws.data_ptr(), tw_w.data_ptr(), tw_h.data_ptr(), _stream()), 'rd_ram_amp')
    return out


def low_freq_mutate_gpu(amp_src, amp_trg, lam, ratio=0.1):
    """The window lerp of fundus.py:21-39 with the mix ratio `lam` given; float32 [C,H,W] device tensor."""
    dev = _dev()
    a = torch.as_tensor(amp_src, dtype=torch.float32).to(dev).contiguous()
    t = torch.as_tensor(amp_trg, dtype=torch.float32).to(dev).contiguous()
    assert a.shape == t.shape and a.dim() == 3
    Cc, H, W = a.shape
    out = torch.empty_like(a)
    L.check(L.lib().rd_ram_mutate(a.data_ptr(), t.data_ptr(), out.data_ptr(), Cc, H, W, window_half_width(H, W, ratio), float(lam), _stream()),
            'rd_ram_mutate')
    return out


def source_to_target_freq_gpu(src_hwc, amp_trg_chw, lam, ratio=0.1):
    """real(ifft2(A' e^{jP})) of fundus.py:41-61 for one HWC image and the partner's amplitude array; no clipping, no
    scaling (the call sites do those); float32 [H,W,3] device tensor."""
    dev = _dev()
    src = torch.as_tensor(src_hwc, dtype=torch.float32).to(dev).contiguous()[None]
    amp = torch.as_tensor(amp_trg_chw, dtype=torch.float32).to(dev).contiguous()[None]
    _, H, W, _ = src.shape
    m = RamMixer(1, H, W, torch.float32, dev, 'fundus', ratio=ratio)
    m.p.clip_lo, m.p.clip_hi, m.p.scale, m.p.div, m.p.offset = -3.0e38, 3.0e38, 1.0, 0.0, 0.0
    oi = torch.empty(1, H, W, 3, device=dev)
    of = torch.empty(1, H, W, 3, device=dev)
    m.bind(src, None, torch.tensor([float(lam)], dtype=torch.float32, device=dev), oi, of, trg_amp=amp)
    m.run()
    return of[0]
