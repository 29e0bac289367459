"""-m gpu: rd_conv_bwd_fused (csrc/conv_fused.hip) -- dgrad + weight gradient of a small-channel 3x3 conv in one launch --
against torch autograd (fp32 math on bf16-representable inputs) and against the separate rd_conv + rd_wgrad launches it
replaces (cudnn dgrad / wgrad behind nn.Conv2d, code/networks/unet.py:37-43,81-88,124-131,281,307).  Through the C ABI."""
import ctypes as C
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from ramdsir import _lib as L                      # noqa: E402
import gpu_util as U                               # noqa: E402
from test_gpu_ops import _conv_desc, _params       # noqa: E402

DT = 'bf16'

# name, [(input mode, Cd, has_norm)], Cout_fwd, dz channel stride (0: = Cout), N, H, W, gstart, BNBWD dz, accumulate, cu_limit
CASES = [
    ('affact32_32_multi_tile', [(L.SRC_AFFACT, 32, True)], 32, 0, 2, 20, 70, [0, 1, 2], 1, 0, 0),
    ('cat16_16_to_32', [(L.SRC_AFFACT, 16, True), (L.SRC_AFFACT, 16, True)], 32, 0, 3, 17, 40, [0, 1, 3], 1, 0, 0),
    ('aff16_16_no_activation_acc', [(L.SRC_AFF, 16, True)], 16, 0, 3, 9, 33, [0, 2, 3], 1, 1, 0),
    ('raw16_32_pooled_input', [(L.SRC_RAW, 16, False)], 32, 0, 2, 24, 32, [0, 1, 2], 1, 0, 0),
    ('affact32_2_from_padded_dlogits', [(L.SRC_AFFACT, 32, True)], 2, 8, 2, 16, 33, [0, 1, 2], 0, 0, 0),
    ('affact16_3_from_padded_dlogits', [(L.SRC_AFFACT, 16, True)], 3, 8, 3, 8, 40, [0, 1, 2, 3], 0, 0, 0),
    ('affact32_16_stored_dz', [(L.SRC_AFFACT, 32, True)], 16, 0, 2, 40, 40, [0, 2], 0, 0, 0),
    ('affact16_16_many_tiles_cu_limit', [(L.SRC_AFFACT, 16, True)], 16, 0, 4, 64, 96, [0, 2, 4], 1, 0, 8),
    ('affact32_32_leaky', [(L.SRC_AFFACT, 32, True)], 32, 0, 2, 8, 32, [0, 1, 2], 1, 0, 0),
]


def _build(case):
    name, in_spec, Cout, dzs, N, H, W, gstart, bnbwd, accumulate, cu_limit = case
    slope = 0.01 if 'leaky' in name else 0.0
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000 + 3)
    keep = U.Keep()
    G = len(gstart) - 1
    ys, virt, prod = [], [], []
    for mode, Cd, has_norm in in_spec:
        z = U.rnd(torch.randn(N, Cd, H, W, generator=gen), DT)
        sc, sh = _params(G, Cd, gen)
        if has_norm:
            y = (z * U.group_rows(sc, gstart, N) + U.group_rows(sh, gstart, N)).requires_grad_(True)
        else:
            y = z.clone().requires_grad_(True)
        a = U.act(y, slope) if mode == L.SRC_AFFACT else y
        ys.append(y); virt.append(a); prod.append((z, sc, sh))
    a = torch.cat(virt, 1)
    Cin = a.shape[1]
    w = U.rnd(torch.randn(Cout, Cin, 3, 3, generator=gen) / np.sqrt(Cin * 9), DT)
    w.requires_grad_(True)
    gz = U.rnd(torch.randn(N, Cout, H, W, generator=gen), DT)
    if bnbwd:
        zc = U.rnd(torch.randn(N, Cout, H, W, generator=gen), DT)
        P, R = _params(G, Cout, gen)
        Q = 0.1 * torch.randn(G, Cout, generator=gen)
        dz = gz * U.group_rows(P, gstart, N) + zc * U.group_rows(Q, gstart, N) + U.group_rows(R, gstart, N)
        src = U.make_src(keep, gz, L.SRC_BNBWD, DT, scale=P, shift=R, ptr2=zc, q=Q)
    else:
        dz = gz
        if dzs:                                            # narrow gradient stored with a zero-padded channel tail (dlogits)
            padded = torch.zeros(N, dzs, H, W)
            padded[:, :Cout] = gz
            src = U.make_src(keep, padded, L.SRC_RAW, DT)
        else:
            src = U.make_src(keep, gz, L.SRC_RAW, DT)
    (F.conv2d(a, w, None, padding=1) * dz).sum().backward()

    def descriptors():
        p = _conv_desc(keep, [src], w.detach(), None, N, H, W, gstart, DT, 9, transpose=True)
        p.emode = 1
        p.c_split = in_spec[0][1] if len(in_spec) == 2 else Cin
        p.cu_limit = cu_limit
        wg = L.RdWgrad()
        outs = []
        for i, (mode, Cd, has_norm) in enumerate(in_spec):
            z, sc, sh = prod[i]
            zd = keep(U.nhwc(z, DT))
            scd, shd = keep(U.fdev(sc)), keep(U.fdev(sh))
            old = U.rnd(torch.randn(ys[i].shape, generator=torch.Generator().manual_seed(77 + i)), DT) if accumulate else torch.zeros(ys[i].shape)
            gbuf = keep(U.nhwc(old if accumulate else torch.full(ys[i].shape, float('nan')), DT))
            bst = keep(torch.zeros(G, L.STAT_SLOTS, Cd, 2, dtype=torch.float64, device=U.dev()))
            d = L.RdDst()
            d.g = gbuf.data_ptr()
            if has_norm:
                d.z, d.scale, d.shift, d.bstats = zd.data_ptr(), scd.data_ptr(), shd.data_ptr(), bst.data_ptr()
            d.kind, d.act, d.accumulate, d.Cd, d.slope, d.n_off, d.g_fixed = L.DST_PLAIN, int(mode == L.SRC_AFFACT), accumulate, Cd, slope, 0, -1
            p.dst[i] = d
            s = L.RdSrc()
            s.ptr = zd.data_ptr()
            if has_norm:
                s.scale, s.shift = scd.data_ptr(), shd.data_ptr()
            s.mode, s.C, s.slope, s.n_off, s.g_fixed = mode, Cd, slope, 0, -1
            wg.a[i] = s
            outs.append((gbuf, bst, old))
        if len(in_spec) == 1:
            p.dst[1].kind = L.DST_NONE
        wg.na, wg.taps, wg.dz = len(in_spec), 9, src
        wg.N, wg.H, wg.W, wg.Cin, wg.Cout, wg.G = N, H, W, Cin, Cout, G
        wg.gstart = L.gstart_array(gstart)
        dW = keep(torch.full((Cout, Cin, 3, 3), float('nan'), device=U.dev()))
        wg.dW, wg.beta = dW.data_ptr(), 0.0
        return p, wg, outs, dW
    return name, keep, descriptors, ys, prod, w, gstart, G


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_fused_backward_matches_autograd_and_the_separate_launches(case):
    name, keep, descriptors, ys, prod, w, gstart, G = _build(case)
    lib = L.lib()
    dt = U.DT[DT][0]
    # ---- fused
    p, wg, outs, dW = descriptors()
    assert lib.rd_conv_bwd_fused_ok(C.byref(p), C.byref(wg), dt) == 1, name
    part = keep(torch.full((lib.rd_conv_bwd_fused_workspace(C.byref(p), C.byref(wg), dt) // 4,), float('nan'), device=U.dev()))
    wg.partial = part.data_ptr()
    L.check(lib.rd_conv_bwd_fused(C.byref(p), C.byref(wg), dt, None), name)
    L.check(lib.rd_conv_bwd_fused_reduce(C.byref(p), C.byref(wg), dt, None), name + ' reduce')
    torch.cuda.synchronize()
    # ---- separate launches on fresh buffers
    p2, wg2, outs2, dW2 = descriptors()
    L.check(lib.rd_conv(C.byref(p2), dt, None), name + ' dgrad')
    part2 = keep(torch.empty(lib.rd_wgrad_workspace(C.byref(wg2), dt) // 4 + 1, device=U.dev()))
    wg2.partial = part2.data_ptr()
    L.check(lib.rd_wgrad(C.byref(wg2), dt, None), name + ' wgrad')
    torch.cuda.synchronize()
    for i in range(len(ys)):
        gbuf, bst, old = outs[i]
        gref = ys[i].grad
        U.assert_close(U.from_nhwc(gbuf), gref + old, DT, '%s.dst%d' % (name, i), scale=2.0)
        # the input gradient is the same arithmetic as conv_small_kernel's: bit for bit
        assert torch.equal(gbuf, outs2[i][0]), '%s.dst%d differs from rd_conv' % (name, i)
        if prod[i] is not None and case[1][i][2]:
            z = prod[i][0]
            rs = torch.stack([torch.stack([gref[gstart[g]:gstart[g + 1]].sum((0, 2, 3)),
                                           (gref * z)[gstart[g]:gstart[g + 1]].sum((0, 2, 3))], -1) for g in range(G)])
            Cd = case[1][i][1]
            tol = 3e-2 * float(rs.abs().max() + gref.abs().sum() / Cd / G * 0.05 + 1e-3)
            assert float((bst.sum(1).cpu().float() - rs).abs().max()) <= tol, '%s bstats' % name
            np.testing.assert_allclose(bst.sum(1).cpu().numpy(), outs2[i][1].sum(1).cpu().numpy(), rtol=1e-4,
                                       atol=1e-4 * float(rs.abs().max()))
    U.assert_close(dW.cpu(), w.grad, DT, name + ' dW')
    # against the stand-alone weight-gradient kernels: same operands, another summation order
    ref_rms = float(w.grad.pow(2).mean().sqrt())
    assert float((dW.cpu() - dW2.cpu()).abs().max()) <= 2e-3 * ref_rms + 1e-6, name


def test_fused_eligibility_rejects_what_the_kernel_does_not_cover():
    case = CASES[0]
    name, keep, descriptors, *_ = _build(case)
    lib = L.lib()
    p, wg, _, _ = descriptors()
    assert lib.rd_conv_bwd_fused_ok(C.byref(p), C.byref(wg), U.DT['bf16'][0]) == 1
    assert lib.rd_conv_bwd_fused_ok(C.byref(p), C.byref(wg), U.DT['f32'][0]) == 0          # bf16 kernels only
    p.dst[0].kind = L.DST_POOL
    assert lib.rd_conv_bwd_fused_ok(C.byref(p), C.byref(wg), U.DT['bf16'][0]) == 0
    p.dst[0].kind = L.DST_PLAIN
    wg.a[0].mode = L.SRC_UP
    assert lib.rd_conv_bwd_fused_ok(C.byref(p), C.byref(wg), U.DT['bf16'][0]) == 0
    wg.a[0].mode = L.SRC_AFFACT
    wg.H += 1
    assert lib.rd_conv_bwd_fused_ok(C.byref(p), C.byref(wg), U.DT['bf16'][0]) == 0
    wg.H -= 1
    assert lib.rd_conv_bwd_fused(C.byref(p), C.byref(wg), U.DT['bf16'][0], None) == -1      # no workspace bound: refused, not a crash
