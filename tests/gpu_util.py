"""Helpers for the -m gpu parity tests: NHWC device tensors, packed weights, descriptors, tolerances."""
import ctypes as C

import numpy as np
import torch
import torch.nn.functional as F

from ramdsir import _lib as L

DT = {'f32': (L.RD_F32, torch.float32), 'bf16': (L.RD_BF16, torch.bfloat16)}
# per-op tolerances relative to the RMS of the reference tensor: max |diff| <= 4 * RTOL * rms (fp32: 8e-5) AND the RMS of the
# element errors <= RMS_TOL * rms (fp32: 1e-5 -- BASELINE.md 3.5's "fp32 <= 1e-5 rel per op" is this bound; the max bound is wider
# because the largest of ~10^6 rounding errors of a K ~ 10^3 accumulation sits several sigma out)
RTOL = {'f32': 2e-5, 'bf16': 1e-2}
RMS_TOL = {'f32': 1e-5, 'bf16': 5e-3}


def dev():
    return torch.device('cuda:0')


def rnd(t, dtype):
    """Round to the storage dtype and back to fp32 (so both sides start from representable values)."""
    return t.to(DT[dtype][1]).float()


def nhwc(t_nchw, dtype):
    return t_nchw.permute(0, 2, 3, 1).contiguous().to(DT[dtype][1]).to(dev())


def from_nhwc(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def fdev(t):
    return None if t is None else t.float().contiguous().to(dev())


def pack_weights(w_oihw, dtype, transpose=False):
    lib = L.lib()
    Cout, Cin, kh, kw = w_oihw.shape
    taps = kh * kw
    n = lib.rd_packed_elems(Cout, Cin, taps, int(transpose), DT[dtype][0])
    out = torch.zeros(n, dtype=DT[dtype][1], device=dev())
    wd = fdev(w_oihw)
    L.check(lib.rd_pack_weights(L.ptr(wd), L.ptr(out), Cout, Cin, taps, int(transpose), DT[dtype][0], None), 'pack')
    torch.cuda.synchronize()
    return out


def pads(Cout, Cin, dtype):
    ck = 32 if dtype == 'bf16' else 16
    r32 = lambda v: (v + 31) // 32 * 32
    rck = lambda v: (v + ck - 1) // ck * ck
    return rck(Cin), r32(Cout)


def assert_close(got, ref, dtype, name='', scale=1.0):
    ref = ref.float()
    got = got.float()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    rms = float(ref.pow(2).mean().sqrt()) + 1e-20
    err = float((got - ref).abs().max())
    assert np.isfinite(err), name
    assert err <= RTOL[dtype] * scale * rms * 4 + 1e-30, '%s: max|diff| %.3e vs rms %.3e (dtype %s)' % (name, err, rms, dtype)
    rms_err = float((got - ref).double().pow(2).mean().sqrt())
    assert rms_err <= RMS_TOL[dtype] * scale * rms + 1e-30, '%s: rms(diff) %.3e vs rms %.3e (dtype %s)' % (name, rms_err, rms, dtype)


def group_rows(param_gc, gstart, N):
    """[G][C] per-group rows -> [N][C][1][1] per-image rows."""
    G = len(gstart) - 1
    rows = []
    for g in range(G):
        rows += [param_gc[g]] * (gstart[g + 1] - gstart[g])
    return torch.stack(rows)[:, :, None, None]


def act(x, slope):
    return F.relu(x) if slope == 0 else F.leaky_relu(x, slope)


def virtual_input(x, mode, scale, shift, slope, gstart):
    """The conv-input tensor a source descriptor denotes, as torch ops (NCHW fp32)."""
    N = x.shape[0]
    if mode == L.SRC_RAW:
        return x
    sc, sh = group_rows(scale, gstart, N), group_rows(shift, gstart, N)
    if mode == L.SRC_AFF:
        return x * sc + sh
    if mode == L.SRC_AFFACT:
        return act(x * sc + sh, slope)
    if mode == L.SRC_POOL:
        return F.max_pool2d(act(x * sc + sh, slope), 2)
    if mode == L.SRC_UP:
        return act(F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False) * sc + sh, slope)
    raise ValueError(mode)


class Keep:
    """Keeps device tensors alive for the lifetime of a descriptor."""
    def __init__(self):
        self.t = []

    def __call__(self, t):
        self.t.append(t)
        return t


def make_src(keep, x_nchw, mode, dtype, scale=None, shift=None, slope=0.0, ptr2=None, q=None, n_off=0, g_fixed=-1):
    s = L.RdSrc()
    xd = keep(nhwc(x_nchw, dtype))
    s.ptr = xd.data_ptr()
    s.ptr2 = keep(nhwc(ptr2, dtype)).data_ptr() if ptr2 is not None else None
    s.scale = keep(fdev(scale)).data_ptr() if scale is not None else None
    s.shift = keep(fdev(shift)).data_ptr() if shift is not None else None
    s.q = keep(fdev(q)).data_ptr() if q is not None else None
    s.mode, s.C, s.slope, s.n_off, s.g_fixed = mode, x_nchw.shape[1], slope, n_off, g_fixed
    return s
