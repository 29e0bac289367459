"""-m gpu: every HIP kernel, called through the C ABI, against a plain torch fp32 restatement of the
same op on the same seeded inputs.  Tolerance = 4 x gpu_util.RTOL x RMS(reference): fp32 8e-5 x RMS (the f32 MFMA is a
bit-exact fmaf chain; only the summation order differs), bf16 4e-2 x RMS (inputs pre-rounded to bf16, operands of
the MFMA and the stored result rounded once each: a few bf16 ulps of the largest elements)."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from ramdsir import _lib as L          # noqa: E402
import gpu_util as U                   # noqa: E402

DTYPES = ['f32', 'bf16']


def _slotted(t_gc2):
    """[G][C][2] sums spread over the RD_STAT_SLOTS copies (uneven on purpose)."""
    G, Cc, _ = t_gc2.shape
    w = torch.rand(L.STAT_SLOTS)
    w = w / w.sum()
    return (t_gc2[:, None].double() * w[None, :, None, None].double())


def _params(G, Cc, gen):
    return 1.0 + 0.3 * torch.randn(G, Cc, generator=gen), 0.2 * torch.randn(G, Cc, generator=gen)


def _conv_desc(keep, srcs, w, bias, N, H, W, gstart, dtype, taps, transpose=False):
    p = L.RdConv()
    for i, s in enumerate(srcs):
        p.src[i] = s
    p.nsrc, p.taps = len(srcs), taps
    wp = keep(U.pack_weights(w, dtype, transpose))
    p.w = wp.data_ptr()
    Cout, Cin = (w.shape[1], w.shape[0]) if transpose else (w.shape[0], w.shape[1])
    p.bias = keep(U.fdev(bias)).data_ptr() if bias is not None else None
    p.CinPad, p.CoutPad = U.pads(Cout, Cin, dtype)
    p.N, p.H, p.W, p.Cin, p.Cout = N, H, W, Cin, Cout
    p.G = len(gstart) - 1
    p.gstart = L.gstart_array(gstart)
    return p


# ------------------------------------------------------------------------------------ conv forward
FWD_CASES = [
    # name, taps, [(mode, C)], Cout, N, H, W, gstart, slope
    ('img3_16', 9, [(L.SRC_RAW, 3)], 16, 2, 20, 36, [0, 2], 0.0),
    ('aff16_16', 9, [(L.SRC_AFF, 16)], 16, 3, 16, 32, [0, 1, 3], 0.0),
    ('affact32_64', 9, [(L.SRC_AFFACT, 32)], 64, 2, 25, 25, [0, 1, 2], 0.0),
    ('leaky64_32', 9, [(L.SRC_AFFACT, 64)], 32, 2, 9, 40, [0, 2], 0.01),
    ('pool16_32', 9, [(L.SRC_POOL, 16)], 32, 2, 12, 20, [0, 1, 2], 0.0),
    ('up32_32', 9, [(L.SRC_UP, 32)], 32, 2, 16, 24, [0, 2], 0.0),
    ('cat16_up16_32', 9, [(L.SRC_AFFACT, 16), (L.SRC_UP, 16)], 32, 2, 16, 40, [0, 1, 2], 0.0),
    ('cat64_up64_128', 9, [(L.SRC_AFFACT, 64), (L.SRC_UP, 64)], 128, 2, 10, 12, [0, 1, 2], 0.0),
    # concat of two plain sources (the stored-upsample form of ConvU.conv3): prefetching / chunk-pipelined loaders
    ('cat16_aff16_16', 9, [(L.SRC_AFFACT, 16), (L.SRC_AFFACT, 16)], 16, 2, 16, 40, [0, 1, 2], 0.0),
    ('cat64_aff64_128', 9, [(L.SRC_AFFACT, 64), (L.SRC_AFFACT, 64)], 128, 3, 10, 12, [0, 1, 3], 0.0),
    ('raw128_64', 9, [(L.SRC_RAW, 128)], 64, 2, 9, 33, [0, 1, 2], 0.0),
    ('out32_2', 9, [(L.SRC_AFFACT, 32)], 2, 2, 16, 33, [0, 1, 2], 0.0),
    ('out16_3', 9, [(L.SRC_AFFACT, 16)], 3, 3, 8, 32, [0, 1, 2, 3], 0.0),
    # conv_small_fwd_kernel: many tiles per image (ragged right / bottom), 1 / 3-of-4 / 4 live channel slots
    ('affact32_32_70x100', 9, [(L.SRC_AFFACT, 32)], 32, 2, 70, 100, [0, 1, 2], 0.0),
    ('raw8_32_45x70', 9, [(L.SRC_RAW, 8)], 32, 2, 45, 70, [0, 2], 0.0),
    ('aff24_16_33x65', 9, [(L.SRC_AFF, 24)], 16, 2, 33, 65, [0, 1, 2], 0.0),
    ('cat16_aff8_32_41x64', 9, [(L.SRC_AFFACT, 16), (L.SRC_AFFACT, 8)], 32, 2, 41, 64, [0, 2], 0.0),
    # ... and its narrow-output variant (<= 4 channels: scalar stores, lanes outside the image write to the trash record)
    ('out32_2_70x100', 9, [(L.SRC_AFFACT, 32)], 2, 2, 70, 100, [0, 1, 2], 0.0),
    ('out16_3_45x70', 9, [(L.SRC_AFFACT, 16)], 3, 3, 45, 70, [0, 1, 2, 3], 0.0),
    ('out8_1_33x65', 9, [(L.SRC_RAW, 8)], 1, 2, 33, 65, [0, 2], 0.0),
    ('out32_4_41x64', 9, [(L.SRC_AFF, 32)], 4, 2, 41, 64, [0, 1, 2], 0.0),
    ('c256_256', 9, [(L.SRC_AFFACT, 256)], 256, 2, 6, 7, [0, 1, 2], 0.0),
    ('c64_64_50x50', 9, [(L.SRC_AFFACT, 64)], 64, 2, 50, 50, [0, 1, 2], 0.0),          # 10 x 25 tiles: 5 x 2 per image, every lane but 6 live
    ('c128_64_23x100', 9, [(L.SRC_AFFACT, 128)], 64, 2, 23, 100, [0, 2], 0.0),          # ... a ragged last tile row
    ('k1_128_64', 1, [(L.SRC_AFFACT, 128)], 64, 2, 10, 34, [0, 1, 2], 0.0),
    ('k1_16_16', 1, [(L.SRC_AFFACT, 16)], 16, 2, 20, 20, [0, 2], 0.0),
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', FWD_CASES, ids=[c[0] for c in FWD_CASES])
def test_conv_forward(case, dtype):
    name, taps, src_spec, Cout, N, H, W, gstart, slope = case
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
    keep = U.Keep()
    G = len(gstart) - 1
    srcs, virt = [], []
    for mode, Cc in src_spec:
        hs, ws = (2 * H, 2 * W) if mode == L.SRC_POOL else ((H // 2, W // 2) if mode == L.SRC_UP else (H, W))
        x = U.rnd(torch.randn(N, Cc, hs, ws, generator=gen), dtype)
        sc, sh = _params(G, Cc, gen)
        srcs.append(U.make_src(keep, x, mode, dtype, sc, sh, slope))
        virt.append(U.virtual_input(x, mode, sc, sh, slope, gstart))
    a = torch.cat(virt, 1)
    Cin = a.shape[1]
    k = 3 if taps == 9 else 1
    w = U.rnd(torch.randn(Cout, Cin, k, k, generator=gen) / np.sqrt(Cin * taps), dtype)
    bias = 0.1 * torch.randn(Cout, generator=gen)
    ref = F.conv2d(a, w, bias, padding=k // 2)
    p = _conv_desc(keep, srcs, w, bias, N, H, W, gstart, dtype, taps)
    out = torch.full((N, H, W, Cout), float('nan'), dtype=U.DT[dtype][1], device=U.dev())
    stats = torch.zeros(G, L.STAT_SLOTS, Cout, 2, dtype=torch.float64, device=U.dev())
    p.emode, p.out, p.stats = 0, out.data_ptr(), stats.data_ptr()
    L.check(L.lib().rd_conv(C.byref(p), U.DT[dtype][0], None), name)
    torch.cuda.synchronize()
    U.assert_close(U.from_nhwc(out), ref, dtype, name)
    nb_ = ref - bias[None, :, None, None]                 # the sums are those of the result WITHOUT the bias (ramdsir.h, RD_STAT_SLOTS)
    ref_stats = torch.stack([torch.stack([nb_[gstart[g]:gstart[g + 1]].sum((0, 2, 3)),
                                          nb_[gstart[g]:gstart[g + 1]].pow(2).sum((0, 2, 3))], -1) for g in range(G)])
    npx = H * W * max(gstart[g + 1] - gstart[g] for g in range(G))
    # sums of npx values: compare relative to sqrt(npx)*rms (stats are taken from the fp32 accumulators)
    assert float((stats.sum(1).cpu().float() - ref_stats).abs().max()) <= (1e-3 if dtype == 'bf16' else 1e-4) * npx * float(ref.abs().max() + 1) ** 2


# ------------------------------------------------------------------------------------ conv gradient (dgrad + epilogues)
GRAD_CASES = [
    # name, taps, [(kind, Cd, act)], Cout_fwd, N, H, W, gstart, slope, accumulate
    ('plain32', 9, [(L.DST_PLAIN, 32, 1)], 32, 2, 12, 36, [0, 1, 2], 0.0, 0),
    ('plain16_noact_acc', 9, [(L.DST_PLAIN, 16, 0)], 16, 3, 9, 20, [0, 1, 3], 0.0, 1),
    ('pool16_acc', 9, [(L.DST_POOL, 16, 1)], 32, 2, 8, 20, [0, 1, 2], 0.0, 1),
    ('pool64_leaky', 9, [(L.DST_POOL, 64, 1)], 64, 2, 5, 6, [0, 2], 0.01, 0),
    ('upy32', 9, [(L.DST_UPY, 32, 1)], 32, 2, 12, 16, [0, 1, 2], 0.0, 0),
    ('cat16_upy16', 9, [(L.DST_PLAIN, 16, 1), (L.DST_UPY, 16, 1)], 32, 2, 16, 40, [0, 1, 2], 0.0, 0),
    ('cat64_upy64', 9, [(L.DST_PLAIN, 64, 1), (L.DST_UPY, 64, 1)], 128, 2, 10, 12, [0, 1, 2], 0.0, 0),
    ('cat16_plain16', 9, [(L.DST_PLAIN, 16, 1), (L.DST_PLAIN, 16, 1)], 16, 2, 16, 40, [0, 1, 2], 0.0, 0),
    ('cat64_plain64_acc', 9, [(L.DST_PLAIN, 64, 1), (L.DST_PLAIN, 64, 1)], 128, 2, 10, 12, [0, 1, 2], 0.0, 1),
    ('k1_plain128', 1, [(L.DST_PLAIN, 128, 1)], 64, 2, 10, 34, [0, 1, 2], 0.0, 0),
    ('from_out2', 9, [(L.DST_PLAIN, 32, 1)], 2, 2, 16, 33, [0, 1, 2], 0.0, 0),
    ('plain64_50x50_acc', 9, [(L.DST_PLAIN, 64, 1)], 64, 2, 50, 50, [0, 1, 2], 0.0, 1),             # 10 x 25 tiles
    ('cat64_plain64_27x100', 9, [(L.DST_PLAIN, 64, 1), (L.DST_PLAIN, 64, 1)], 128, 2, 27, 100, [0, 2], 0.0, 0),
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', GRAD_CASES, ids=[c[0] for c in GRAD_CASES])
def test_conv_gradient_epilogues(case, dtype):
    _run_gradient_case(case, dtype, False)


ROW_BLOCK_CASES = [
    # one input chunk (<= 32 channels of dz) -> 33..64 channels of one plain tensor, launched as the engine launches dec.convu1.conv1's
    # dgrad: two launches over the 32-row blocks of the packed weights (rd_conv_t.w_tap_rows)
    ('plain64_from32_40x70', 9, [(L.DST_PLAIN, 64, 1)], 32, 2, 40, 70, [0, 1, 2], 0.0, 0),
    ('plain48_from24_33x50_acc', 9, [(L.DST_PLAIN, 48, 1)], 24, 2, 33, 50, [0, 2], 0.0, 1),
    ('plain64_noact_from16_20x36', 9, [(L.DST_PLAIN, 64, 0)], 16, 3, 20, 36, [0, 1, 3], 0.0, 0),
    ('k1_plain64_from32_30x44', 1, [(L.DST_PLAIN, 64, 1)], 32, 2, 30, 44, [0, 1, 2], 0.0, 0),
]


@pytest.mark.parametrize('halves', [False, True], ids=['one_launch', 'row_blocks'])
@pytest.mark.parametrize('case', ROW_BLOCK_CASES, ids=[c[0] for c in ROW_BLOCK_CASES])
def test_conv_gradient_over_row_blocks_of_the_packed_weights(case, halves):
    _run_gradient_case(case, 'bf16', halves)


def _run_gradient_case(case, dtype, halves):
    name, taps, dst_spec, Cout, N, H, W, gstart, slope, accumulate = case
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000 + 1)
    keep = U.Keep()
    G = len(gstart) - 1
    k = 3 if taps == 9 else 1
    ys, virt, prod = [], [], []
    for kind, Cd, act in dst_spec:
        hs, ws = (2 * H, 2 * W) if kind == L.DST_POOL else ((H // 2, W // 2) if kind == L.DST_UPY else (H, W))
        z = U.rnd(torch.randn(N, Cd, hs, ws, generator=gen), dtype)
        sc, sh = _params(G, Cd, gen)
        scn, shn = U.group_rows(sc, gstart, N), U.group_rows(sh, gstart, N)
        zz = F.interpolate(z, scale_factor=2, mode='bilinear', align_corners=False) if kind == L.DST_UPY else z
        y = (zz * scn + shn).requires_grad_(True)
        a = U.act(y, slope) if act else y
        if kind == L.DST_POOL:
            a = F.max_pool2d(a, 2)
        ys.append(y)
        virt.append(a)
        prod.append((z, zz, sc, sh))
    a = torch.cat(virt, 1)
    Cin = a.shape[1]
    w = U.rnd(torch.randn(Cout, Cin, k, k, generator=gen) / np.sqrt(Cin * taps), dtype)
    dz = U.rnd(torch.randn(N, Cout, H, W, generator=gen), dtype)
    (F.conv2d(a, w, None, padding=k // 2) * dz).sum().backward()
    src = U.make_src(keep, dz, L.SRC_RAW, dtype)
    p = _conv_desc(keep, [src], w, None, N, H, W, gstart, dtype, taps, transpose=True)
    p.emode = 1
    p.c_split = dst_spec[0][1] if len(dst_spec) == 2 else Cin
    outs = []
    for i, (kind, Cd, act) in enumerate(dst_spec):
        z, zz, sc, sh = prod[i]
        d = L.RdDst()
        old = U.rnd(torch.randn(ys[i].shape, generator=gen), dtype) if accumulate else torch.zeros(ys[i].shape)
        gbuf = keep(U.nhwc(old if accumulate else torch.full(ys[i].shape, float('nan')), dtype))
        bst = keep(torch.zeros(G, L.STAT_SLOTS, Cd, 2, dtype=torch.float64, device=U.dev()))
        d.g, d.z = gbuf.data_ptr(), keep(U.nhwc(z, dtype)).data_ptr()
        d.scale, d.shift = keep(U.fdev(sc)).data_ptr(), keep(U.fdev(sh)).data_ptr()
        d.bstats, d.kind, d.act, d.accumulate, d.Cd, d.slope, d.n_off, d.g_fixed = bst.data_ptr(), kind, act, accumulate, Cd, slope, 0, -1
        p.dst[i] = d
        outs.append((gbuf, bst, old))
    if len(dst_spec) == 1:
        p.dst[1].kind = L.DST_NONE
    if halves:
        assert len(dst_spec) == 1 and p.CinPad == 32 and p.CoutPad == 64
        d0 = p.dst[0]
        for kb in range(2):
            q = L.RdConv()
            C.memmove(C.byref(q), C.byref(p), C.sizeof(L.RdConv))
            q.Cout = q.c_split = min(32, p.Cout - 32 * kb)
            q.CoutPad, q.w_tap_rows = 32, p.CoutPad
            q.w = p.w + 32 * kb * p.CinPad * 2
            q.dst[0].g, q.dst[0].z = d0.g + 64 * kb, d0.z + 64 * kb
            q.dst[0].scale, q.dst[0].shift, q.dst[0].bstats = d0.scale + 128 * kb, d0.shift + 128 * kb, d0.bstats + 512 * kb
            L.check(L.lib().rd_conv(C.byref(q), U.DT[dtype][0], None), name + '.rows%d' % kb)
        bad = L.RdConv()
        C.memmove(C.byref(bad), C.byref(p), C.sizeof(L.RdConv))
        bad.w_tap_rows = 64                                   # only 32-row launches of the small-channel class may carry it
        assert L.lib().rd_conv(C.byref(bad), U.DT[dtype][0], None) == -2
    else:
        L.check(L.lib().rd_conv(C.byref(p), U.DT[dtype][0], None), name)
    torch.cuda.synchronize()
    for i, (kind, Cd, act) in enumerate(dst_spec):
        gbuf, bst, old = outs[i]
        gref = ys[i].grad
        U.assert_close(U.from_nhwc(gbuf), gref + old, dtype, '%s.dst%d' % (name, i), scale=2.0)
        zz = prod[i][1]
        rs = torch.stack([torch.stack([gref[gstart[g]:gstart[g + 1]].sum((0, 2, 3)),
                                       (gref * zz)[gstart[g]:gstart[g + 1]].sum((0, 2, 3))], -1) for g in range(G)])
        tol = (3e-2 if dtype == 'bf16' else 2e-4) * float(rs.abs().max() + gref.abs().sum() / Cd / G * 0.05 + 1e-3)
        assert float((bst.sum(1).cpu().float() - rs).abs().max()) <= tol, '%s bstats' % name


@pytest.mark.parametrize('dtype', DTYPES)
def test_conv_bnbwd_loader_and_image_offsets(dtype):
    """dz = P*g + Q*z + R folded into the read; n_off / g_fixed as the rec decoder uses them on x5."""
    gen = torch.Generator().manual_seed(11)
    keep = U.Keep()
    N, H, W, Cz, Ca = 2, 9, 20, 32, 64
    gstart = [0, 1, 2]
    g = U.rnd(torch.randn(N, Cz, H, W, generator=gen), dtype)
    z = U.rnd(torch.randn(N, Cz, H, W, generator=gen), dtype)
    P, R = _params(2, Cz, gen)
    Q = 0.1 * torch.randn(2, Cz, generator=gen)
    dz = g * U.group_rows(P, gstart, N) + z * U.group_rows(Q, gstart, N) + U.group_rows(R, gstart, N)
    w = U.rnd(torch.randn(Cz, Ca, 3, 3, generator=gen) / 24, dtype)
    # destination: a 4-image producer tensor; this launch covers its images 2..3 which live in producer group 1
    zprod = U.rnd(torch.randn(4, Ca, H, W, generator=gen), dtype)
    sc, sh = _params(2, Ca, gen)
    y = (zprod[2:4] * sc[1][None, :, None, None] + sh[1][None, :, None, None]).requires_grad_(True)
    (F.conv2d(F.relu(y), w, None, padding=1) * dz).sum().backward()
    src = U.make_src(keep, g, L.SRC_BNBWD, dtype, scale=P, shift=R, ptr2=z, q=Q)
    p = _conv_desc(keep, [src], w, None, N, H, W, gstart, dtype, 9, transpose=True)
    p.emode, p.c_split = 1, Ca
    old = U.rnd(torch.randn(4, Ca, H, W, generator=gen), dtype)
    gbuf = U.nhwc(old, dtype)
    bst = torch.zeros(2, L.STAT_SLOTS, Ca, 2, dtype=torch.float64, device=U.dev())
    d = L.RdDst()
    d.g, d.z = gbuf.data_ptr(), keep(U.nhwc(zprod, dtype)).data_ptr()
    d.scale, d.shift = keep(U.fdev(sc)).data_ptr(), keep(U.fdev(sh)).data_ptr()
    d.bstats, d.kind, d.act, d.accumulate, d.Cd, d.slope, d.n_off, d.g_fixed = bst.data_ptr(), L.DST_PLAIN, 1, 1, Ca, 0.0, 2, 1
    p.dst[0] = d
    p.dst[1].kind = L.DST_NONE
    L.check(L.lib().rd_conv(C.byref(p), U.DT[dtype][0], None), 'bnbwd')
    torch.cuda.synchronize()
    ref = old.clone()
    ref[2:4] += y.grad
    U.assert_close(U.from_nhwc(gbuf), ref, dtype, 'bnbwd', scale=3.0)
    assert float(bst[0].abs().max()) == 0.0                 # only producer group 1 was touched
    np.testing.assert_allclose(bst.sum(1)[1, :, 0].cpu(), y.grad.sum((0, 2, 3)), rtol=0, atol=(0.5 if dtype == 'bf16' else 5e-3))


def _assert_bf16_rounding_of(got, ref, terms, what):
    """got == round-to-nearest-bf16(ref) up to the fp32 rounding of the expression: |got - ref| <= half a bf16 spacing at |ref| plus a
    few fp32 ulps of the TERMS that were added (an fma chain against separate multiplies and adds; results that cancel to ~0)."""
    m = torch.maximum(got.abs(), ref.abs()).clamp_min(1e-30)
    tol = 0.5 * 2.0 ** (torch.floor(torch.log2(m)) - 7) * 1.001 + 4 * 2.0 ** -24 * terms
    bad = (got - ref).abs() > tol
    assert not bool(bad.any()), '%s: %d values off, worst %.3g' % (what, int(bad.sum()), float(((got - ref).abs() - tol).max()))


@pytest.mark.parametrize('case', ['affact64', 'cat_aff64_affact64'])
def test_conv_forward_also_stores_its_staged_sources(case):
    """rd_src_t.out (ramdsir.h): a forward launch that runs on the warp-specialised 64-wide kernel also writes act(scale x + shift)
    of every source that asks for it -- every pixel once, images and channels of the source's own layout -- and
    rd_conv_honours_src_out() says so beforehand.  The conv result itself is unchanged.  (Geometry large enough for the product
    library's routing: >= 300 workgroups of 64 output channels, ragged right and bottom tiles.)"""
    gen = torch.Generator().manual_seed(5)
    keep = U.Keep()
    N, H, W, Cout, gstart, dtype = 8, 103, 95, 64, [0, 3, 8], 'bf16'
    spec = [(L.SRC_AFFACT, 64, 0.0)] if case == 'affact64' else [(L.SRC_AFF, 64, 0.0), (L.SRC_AFFACT, 64, 0.01)]
    srcs, virt, outs, terms = [], [], [], []
    for mode, Cc, slope in spec:
        x = U.rnd(torch.randn(N, Cc, H, W, generator=gen), dtype)
        sc, sh = _params(2, Cc, gen)
        terms.append((x * U.group_rows(sc, gstart, N)).abs() + U.group_rows(sh, gstart, N).abs())
        src = U.make_src(keep, x, mode, dtype, sc, sh, slope)
        outs.append(torch.full((N, H, W, Cc), float('nan'), dtype=torch.bfloat16, device=U.dev()))
        src.out = outs[-1].data_ptr()
        srcs.append(src)
        virt.append(U.virtual_input(x, mode, sc, sh, slope, gstart))
    a = torch.cat(virt, 1)
    w = U.rnd(torch.randn(Cout, a.shape[1], 3, 3, generator=gen) / np.sqrt(a.shape[1] * 9), dtype)
    bias = 0.1 * torch.randn(Cout, generator=gen)
    p = _conv_desc(keep, srcs, w, bias, N, H, W, gstart, dtype, 9)
    out = torch.full((N, H, W, Cout), float('nan'), dtype=torch.bfloat16, device=U.dev())
    p.emode, p.out, p.stats = 0, out.data_ptr(), None
    if os.environ.get('RAMDSIR_DEBUG_LIB') == '1' and not L.lib().rd_conv_honours_src_out(C.byref(p), L.RD_BF16):
        pytest.skip('forced dispatch routes this launch away from the kernel that stores its sources')
    assert L.lib().rd_conv_honours_src_out(C.byref(p), L.RD_BF16) == 1
    L.check(L.lib().rd_conv(C.byref(p), L.RD_BF16, None), case)
    torch.cuda.synchronize()
    U.assert_close(U.from_nhwc(out), F.conv2d(a, w, bias, padding=1), dtype, case)
    for o, v, t in zip(outs, virt, terms):
        got = U.from_nhwc(o)
        assert torch.isfinite(got).all()                          # every pixel of every image was written
        _assert_bf16_rounding_of(got, v, t, case)
    # a launch that goes to another kernel (32 output channels) says so, and leaves `out` alone
    w32 = U.rnd(torch.randn(32, a.shape[1], 3, 3, generator=gen) / 24, dtype)
    q = _conv_desc(keep, srcs, w32, None, N, H, W, gstart, dtype, 9)
    o32 = torch.zeros((N, H, W, 32), dtype=torch.bfloat16, device=U.dev())
    q.emode, q.out, q.stats = 0, o32.data_ptr(), None
    assert L.lib().rd_conv_honours_src_out(C.byref(q), L.RD_BF16) == 0
    # the 64-wide kernel addresses its destinations with 32-bit offsets built from 24-bit multiplies: a destination of 2^24 pixels or
    # more is routed elsewhere by the same function (descriptor only, nothing is launched)
    n_keep = p.N
    p.N = (1 << 24) // (H * W) + 1
    assert L.lib().rd_conv_honours_src_out(C.byref(p), L.RD_BF16) == 0
    p.N = n_keep
    assert L.lib().rd_conv_honours_src_out(C.byref(p), L.RD_BF16) == 1


def test_conv_gradient_also_stores_the_dz_it_forms():
    """The gradient launch on a BatchNorm-backward pair (g, z) writes dz = P g + Q z + R to rd_src_t.out; the weight gradient of the
    layer reads it as a stored operand (engine.Plan.store_operands)."""
    gen = torch.Generator().manual_seed(12)
    keep = U.Keep()
    N, H, W, Cz, Ca, dtype = 7, 100, 104, 64, 64, 'bf16'
    gstart = [0, 2, 7]
    g = U.rnd(torch.randn(N, Cz, H, W, generator=gen), dtype)
    z = U.rnd(torch.randn(N, Cz, H, W, generator=gen), dtype)
    P, R = _params(2, Cz, gen)
    Q = 0.1 * torch.randn(2, Cz, generator=gen)
    dz = g * U.group_rows(P, gstart, N) + z * U.group_rows(Q, gstart, N) + U.group_rows(R, gstart, N)
    w = U.rnd(torch.randn(Cz, Ca, 3, 3, generator=gen) / 24, dtype)
    zprod = U.rnd(torch.randn(N, Ca, H, W, generator=gen), dtype)
    sc, sh = _params(2, Ca, gen)
    y = (zprod * U.group_rows(sc, gstart, N) + U.group_rows(sh, gstart, N)).requires_grad_(True)
    (F.conv2d(F.relu(y), w, None, padding=1) * dz).sum().backward()
    src = U.make_src(keep, g, L.SRC_BNBWD, dtype, scale=P, shift=R, ptr2=z, q=Q)
    dz_out = torch.full((N, H, W, Cz), float('nan'), dtype=torch.bfloat16, device=U.dev())
    src.out = dz_out.data_ptr()
    p = _conv_desc(keep, [src], w, None, N, H, W, gstart, dtype, 9, transpose=True)
    p.emode, p.c_split = 1, Ca
    gbuf = torch.zeros((N, H, W, Ca), dtype=torch.bfloat16, device=U.dev())
    bst = torch.zeros(2, L.STAT_SLOTS, Ca, 2, dtype=torch.float64, device=U.dev())
    d = L.RdDst()
    d.g, d.z = gbuf.data_ptr(), keep(U.nhwc(zprod, dtype)).data_ptr()
    d.scale, d.shift = keep(U.fdev(sc)).data_ptr(), keep(U.fdev(sh)).data_ptr()
    d.bstats, d.kind, d.act, d.accumulate, d.Cd, d.slope, d.n_off, d.g_fixed = bst.data_ptr(), L.DST_PLAIN, 1, 0, Ca, 0.0, 0, -1
    p.dst[0] = d
    p.dst[1].kind = L.DST_NONE
    if os.environ.get('RAMDSIR_DEBUG_LIB') == '1' and not L.lib().rd_conv_honours_src_out(C.byref(p), L.RD_BF16):
        pytest.skip('forced dispatch routes this launch away from the kernel that stores its sources')
    assert L.lib().rd_conv_honours_src_out(C.byref(p), L.RD_BF16) == 1
    L.check(L.lib().rd_conv(C.byref(p), L.RD_BF16, None), 'dz_out')
    torch.cuda.synchronize()
    U.assert_close(U.from_nhwc(gbuf), y.grad, dtype, 'dz_out dgrad', scale=3.0)
    got = U.from_nhwc(dz_out)
    assert torch.isfinite(got).all()
    _assert_bf16_rounding_of(got, dz, (g * U.group_rows(P, gstart, N)).abs() + (z * U.group_rows(Q, gstart, N)).abs() + U.group_rows(R, gstart, N).abs(), 'dz')
    # the same gradient launch on a STORED dz (plain source): it runs on an instantiation that never writes rd_src_t.out, the query says so
    # beforehand (one predicate for query and dispatch, csrc/conv_pp.hip rd_conv_ws_stores_sources) and the tensor is left alone
    src2 = U.make_src(keep, dz, L.SRC_RAW, dtype)
    untouched = torch.full((N, H, W, Cz), 7.0, dtype=torch.bfloat16, device=U.dev())
    src2.out = untouched.data_ptr()
    p2 = _conv_desc(keep, [src2], w, None, N, H, W, gstart, dtype, 9, transpose=True)
    p2.emode, p2.c_split = 1, Ca
    p2.dst[0] = d
    p2.dst[1].kind = L.DST_NONE
    assert L.lib().rd_conv_honours_src_out(C.byref(p2), L.RD_BF16) == 0
    bst.zero_()
    L.check(L.lib().rd_conv(C.byref(p2), L.RD_BF16, None), 'raw dz with out')
    torch.cuda.synchronize()
    assert bool((untouched == 7.0).all())
    U.assert_close(U.from_nhwc(gbuf), y.grad, dtype, 'raw dz dgrad', scale=3.0)


# ------------------------------------------------------------------------------------ wgrad
WG_CASES = [
    ('img3_16', 9, [(L.SRC_RAW, 3)], 16, 2, 20, 36, 0),
    ('affact32_64', 9, [(L.SRC_AFFACT, 32)], 64, 3, 25, 25, 1),
    ('pool16_32', 9, [(L.SRC_POOL, 16)], 32, 2, 12, 20, 1),
    ('cat16_up16_32', 9, [(L.SRC_AFFACT, 16), (L.SRC_UP, 16)], 32, 2, 16, 40, 1),
    ('c128_128', 9, [(L.SRC_AFFACT, 128)], 128, 2, 10, 12, 1),
    ('cat16_aff16_16', 9, [(L.SRC_AFFACT, 16), (L.SRC_AFFACT, 16)], 16, 2, 16, 40, 1),
    ('cat64_aff64_64_rawdz', 9, [(L.SRC_AFFACT, 64), (L.SRC_AFFACT, 64)], 64, 3, 10, 12, 0),
    ('c64_64_many_tiles', 9, [(L.SRC_AFFACT, 64)], 64, 4, 40, 70, 1),
    ('raw64_64_rawdz_many_tiles', 9, [(L.SRC_RAW, 64)], 64, 4, 40, 70, 0),          # both operands stored (rd_src_t.out): the loader only copies
    ('cat_raw64_raw64_64', 9, [(L.SRC_RAW, 64), (L.SRC_RAW, 64)], 64, 3, 25, 25, 1),
    ('c256_128', 9, [(L.SRC_AFFACT, 256)], 128, 2, 6, 7, 1),
    ('c64_64_25x25', 9, [(L.SRC_AFFACT, 64)], 64, 3, 25, 25, 1),       # 4-row tiles of the warp-specialised kernel: one live row in the last
    ('c128_64_25x25_rawdz', 9, [(L.SRC_AFFACT, 128)], 64, 3, 25, 25, 0),
    # wgrad_sym_kernel (round 6: 128 x 64-channel blocks, Cout % 128 == 0): concatenated input with three groups and a ragged last tile row /
    # column, both operands stored, many tiles per workgroup, gradient channels that do not fill the last 128-block, a wide image border
    ('sym_cat128_128_256', 9, [(L.SRC_AFFACT, 128), (L.SRC_AFFACT, 128)], 256, 3, 25, 25, 1, [0, 1, 2, 3]),
    ('sym_raw128_128_rawdz_many_tiles', 9, [(L.SRC_RAW, 128)], 128, 4, 40, 70, 0),
    ('sym_c64_128_many_tiles', 9, [(L.SRC_AFFACT, 64)], 128, 4, 50, 50, 1),
    ('sym_c128_120_ragged', 9, [(L.SRC_AFFACT, 128)], 120, 2, 13, 34, 1),
    ('sym_aff64_256_100wide', 9, [(L.SRC_AFF, 64)], 256, 2, 9, 100, 0),
    # 64-wide layers (wgrad_ws_kernel; a 64 x 64 form of the symmetric kernel measured slower, docs/experiments.md): a block that straddles the two sources of a concatenated input (dec.convu2.conv3:
    # 32 + 32), a raw source beside a BatchNorm + ReLU one, a ragged image
    ('cat32_32_64_straddle', 9, [(L.SRC_AFFACT, 32), (L.SRC_AFF, 32)], 64, 3, 25, 25, 1, [0, 1, 2, 3]),
    ('cat_raw32_affact32_64_rawdz', 9, [(L.SRC_RAW, 32), (L.SRC_AFFACT, 32)], 64, 2, 13, 70, 0),
    ('c192_56_ragged', 9, [(L.SRC_AFFACT, 192)], 56, 2, 9, 34, 1),
    ('out32_2', 9, [(L.SRC_AFFACT, 32)], 2, 2, 16, 33, 0),
    ('k1_128_64', 1, [(L.SRC_AFFACT, 128)], 64, 2, 10, 34, 0),
    ('k1_16_16', 1, [(L.SRC_AFFACT, 16)], 16, 2, 20, 20, 0),
    # exact geometries of the fused step at 32x32 (3 DSBN groups of 2/3/3 images; 2 groups of 8)
    ('rec_u1c3', 9, [(L.SRC_UP, 16)], 16, 8, 32, 32, 1, [0, 2, 5, 8]),
    ('dec_u1c2', 1, [(L.SRC_AFFACT, 32)], 16, 16, 16, 16, 0, [0, 8, 16]),
    ('dec_u1c1', 9, [(L.SRC_AFFACT, 64)], 32, 16, 16, 16, 1, [0, 8, 16]),
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', WG_CASES, ids=[c[0] for c in WG_CASES])
def test_wgrad(case, dtype):
    name, taps, src_spec, Cout, N, H, W, bnbwd = case[:8]
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000 + 2)
    keep = U.Keep()
    gstart = case[8] if len(case) > 8 else [0, 1, N]
    G = len(gstart) - 1
    k = 3 if taps == 9 else 1
    p = L.RdWgrad()
    virt = []
    for i, (mode, Cc) in enumerate(src_spec):
        hs, ws = (2 * H, 2 * W) if mode == L.SRC_POOL else ((H // 2, W // 2) if mode == L.SRC_UP else (H, W))
        x = U.rnd(torch.randn(N, Cc, hs, ws, generator=gen), dtype)
        sc, sh = _params(G, Cc, gen)
        p.a[i] = U.make_src(keep, x, mode, dtype, sc, sh, 0.0)
        virt.append(U.virtual_input(x, mode, sc, sh, 0.0, gstart))
    a = torch.cat(virt, 1)
    Cin = a.shape[1]
    gz = U.rnd(torch.randn(N, Cout, H, W, generator=gen), dtype)
    if bnbwd:
        z = U.rnd(torch.randn(N, Cout, H, W, generator=gen), dtype)
        P, R = _params(G, Cout, gen)
        Q = 0.1 * torch.randn(G, Cout, generator=gen)
        dz = gz * U.group_rows(P, gstart, N) + z * U.group_rows(Q, gstart, N) + U.group_rows(R, gstart, N)
        p.dz = U.make_src(keep, gz, L.SRC_BNBWD, dtype, scale=P, shift=R, ptr2=z, q=Q)
    else:
        dz = gz
        p.dz = U.make_src(keep, gz, L.SRC_RAW, dtype)
    w = torch.zeros(Cout, Cin, k, k, requires_grad=True)
    (F.conv2d(a, w, None, padding=k // 2) * dz).sum().backward()
    p.na, p.taps, p.N, p.H, p.W, p.Cin, p.Cout, p.G = len(src_spec), taps, N, H, W, Cin, Cout, G
    p.gstart = L.gstart_array(gstart)
    ws_bytes = L.lib().rd_wgrad_workspace(C.byref(p), U.DT[dtype][0])
    part = torch.empty(ws_bytes // 4, device=U.dev())
    old = torch.randn(Cout, Cin, k, k, generator=gen)
    dW = old.clone().to(U.dev())
    p.partial, p.dW, p.beta = part.data_ptr(), dW.data_ptr(), 1.0
    L.check(L.lib().rd_wgrad(C.byref(p), U.DT[dtype][0], None), name)
    torch.cuda.synchronize()
    U.assert_close(dW.cpu() - old, w.grad, dtype, name)


# ------------------------------------------------------------------------------------ padded narrow tensors, rd_bn_apply
@pytest.mark.parametrize('dtype', DTYPES)
def test_padded_narrow_sources(dtype):
    """The 3-channel image and the 2-channel dlogits stored with the channel vector padded to one 16-byte slot
    (descriptor C = slot width, conv Cin = real width): forward, dgrad and wgrad must ignore the pad."""
    S = 8 if dtype == 'bf16' else 4
    gen = torch.Generator().manual_seed(77)
    keep = U.Keep()
    N, H, W, gstart = 2, 20, 36, [0, 1, 2]
    # forward: image 3 -> 16
    x = U.rnd(torch.randn(N, 3, H, W, generator=gen), dtype)
    xpad = torch.cat([x, torch.zeros(N, S - 3, H, W)], 1)
    w = U.rnd(torch.randn(16, 3, 3, 3, generator=gen) / 5, dtype)
    bias = 0.1 * torch.randn(16, generator=gen)
    ref = F.conv2d(x, w, bias, padding=1)
    src = U.make_src(keep, xpad, L.SRC_RAW, dtype)
    p = _conv_desc(keep, [src], w, bias, N, H, W, gstart, dtype, 9)
    out = torch.full((N, H, W, 16), float('nan'), dtype=U.DT[dtype][1], device=U.dev())
    p.emode, p.out, p.stats = 0, out.data_ptr(), None
    L.check(L.lib().rd_conv(C.byref(p), U.DT[dtype][0], None), 'padded fwd')
    torch.cuda.synchronize()
    U.assert_close(U.from_nhwc(out), ref, dtype, 'padded fwd')
    # wgrad of the same conv (a padded) and of out1 (dz padded): 16 -> 2
    dz = U.rnd(torch.randn(N, 16, H, W, generator=gen), dtype)
    wz = torch.zeros(16, 3, 3, 3, requires_grad=True)
    (F.conv2d(x, wz, None, padding=1) * dz).sum().backward()
    g = L.RdWgrad()
    g.a[0], g.dz = src, U.make_src(keep, dz, L.SRC_RAW, dtype)
    g.na, g.taps, g.N, g.H, g.W, g.Cin, g.Cout, g.G = 1, 9, N, H, W, 3, 16, 2
    g.gstart = L.gstart_array(gstart)
    part = torch.empty(L.lib().rd_wgrad_workspace(C.byref(g), U.DT[dtype][0]) // 4, device=U.dev())
    dW = torch.zeros(16, 3, 3, 3, device=U.dev())
    g.partial, g.dW, g.beta = part.data_ptr(), dW.data_ptr(), 0.0
    L.check(L.lib().rd_wgrad(C.byref(g), U.DT[dtype][0], None), 'padded wgrad a')
    torch.cuda.synchronize()
    U.assert_close(dW.cpu(), wz.grad, dtype, 'padded wgrad a')
    a16 = U.rnd(torch.randn(N, 16, H, W, generator=gen), dtype)
    dl = U.rnd(torch.randn(N, 2, H, W, generator=gen), dtype)
    dlpad = torch.cat([dl, torch.zeros(N, S - 2, H, W)], 1)
    w2 = torch.zeros(2, 16, 3, 3, requires_grad=True)
    (F.conv2d(a16, w2, None, padding=1) * dl).sum().backward()
    g2 = L.RdWgrad()
    g2.a[0], g2.dz = U.make_src(keep, a16, L.SRC_RAW, dtype), U.make_src(keep, dlpad, L.SRC_RAW, dtype)
    g2.na, g2.taps, g2.N, g2.H, g2.W, g2.Cin, g2.Cout, g2.G = 1, 9, N, H, W, 16, 2, 2
    g2.gstart = L.gstart_array(gstart)
    part2 = torch.empty(L.lib().rd_wgrad_workspace(C.byref(g2), U.DT[dtype][0]) // 4, device=U.dev())
    dW2 = torch.zeros(2, 16, 3, 3, device=U.dev())
    g2.partial, g2.dW, g2.beta = part2.data_ptr(), dW2.data_ptr(), 0.0
    L.check(L.lib().rd_wgrad(C.byref(g2), U.DT[dtype][0], None), 'padded wgrad dz')
    torch.cuda.synchronize()
    U.assert_close(dW2.cpu(), w2.grad, dtype, 'padded wgrad dz')
    # dgrad from the padded dlogits into a 16-channel producer (no BN: plain gradient)
    wt = U.rnd(torch.randn(2, 16, 3, 3, generator=gen) / 12, dtype)
    yv = a16.clone().requires_grad_(True)
    (F.conv2d(yv, wt, None, padding=1) * dl).sum().backward()
    pd = _conv_desc(keep, [U.make_src(keep, dlpad, L.SRC_RAW, dtype)], wt, None, N, H, W, gstart, dtype, 9, transpose=True)
    pd.emode, pd.c_split = 1, 16
    gbuf = torch.full((N, H, W, 16), float('nan'), dtype=U.DT[dtype][1], device=U.dev())
    d = L.RdDst()
    d.g, d.kind, d.act, d.accumulate, d.Cd, d.slope, d.n_off, d.g_fixed = gbuf.data_ptr(), L.DST_PLAIN, 0, 0, 16, 0.0, 0, -1
    pd.dst[0] = d
    pd.dst[1].kind = L.DST_NONE
    L.check(L.lib().rd_conv(C.byref(pd), U.DT[dtype][0], None), 'padded dgrad')
    torch.cuda.synchronize()
    U.assert_close(U.from_nhwc(gbuf), yv.grad, dtype, 'padded dgrad', scale=2.0)


@pytest.mark.parametrize('dtype', DTYPES)
def test_bn_apply(dtype):
    """out = act(a*x + b*x2 + c): the stored relu(bn(z)) and the stored BN-backward gradient P*g + Q*z + R."""
    gen = torch.Generator().manual_seed(78)
    N, Cc, H, W, gstart = 3, 64, 7, 9, [0, 1, 3]
    x = U.rnd(torch.randn(N, Cc, H, W, generator=gen), dtype)
    z = U.rnd(torch.randn(N, Cc, H, W, generator=gen), dtype)
    a, c = _params(2, Cc, gen)
    b = 0.1 * torch.randn(2, Cc, generator=gen)
    gs = L.gstart_array(gstart)
    xd, zd = U.nhwc(x, dtype), U.nhwc(z, dtype)
    ad, bd, cd = U.fdev(a), U.fdev(b), U.fdev(c)
    out = torch.full((N, H, W, Cc), float('nan'), dtype=U.DT[dtype][1], device=U.dev())
    for slope in (0.0, 0.01):
        L.check(L.lib().rd_bn_apply(L.ptr(xd), None, L.ptr(out), L.ptr(ad), None, L.ptr(cd), slope, N, H, W, Cc, 2, gs,
                                    U.DT[dtype][0], None), 'apply fwd')
        torch.cuda.synchronize()
        ref = U.act(x * U.group_rows(a, gstart, N) + U.group_rows(c, gstart, N), slope)
        U.assert_close(U.from_nhwc(out), ref, dtype, 'bn_apply fwd')
    L.check(L.lib().rd_bn_apply(L.ptr(xd), L.ptr(zd), L.ptr(out), L.ptr(ad), L.ptr(bd), L.ptr(cd), 1.0, N, H, W, Cc, 2, gs,
                                U.DT[dtype][0], None), 'apply bwd')
    torch.cuda.synchronize()
    ref = x * U.group_rows(a, gstart, N) + z * U.group_rows(b, gstart, N) + U.group_rows(c, gstart, N)
    U.assert_close(U.from_nhwc(out), ref, dtype, 'bn_apply bwd')
    assert L.lib().rd_bn_apply(L.ptr(xd), L.ptr(zd), L.ptr(out), L.ptr(ad), None, L.ptr(cd), 1.0, N, H, W, Cc, 2, gs,
                               U.DT[dtype][0], None) != 0           # x2 without its coefficient row: rejected


# ------------------------------------------------------------------------------------ BN finalize
def test_bn_finalize_forward_and_backward():
    gen = torch.Generator().manual_seed(3)
    Cc, gstart = 24, [0, 2, 5]
    G, N, H, W = 2, 5, 6, 7
    x = torch.randn(N, Cc, H, W, generator=gen) * 2 + 0.5
    gam, bet = _params(1, Cc, gen)
    stats = torch.stack([torch.stack([x[gstart[g]:gstart[g + 1]].sum((0, 2, 3)), x[gstart[g]:gstart[g + 1]].pow(2).sum((0, 2, 3))], -1)
                         for g in range(G)])
    stats = _slotted(stats).to(U.dev())
    bufs = {k: torch.zeros(G, Cc, device=U.dev()) for k in ('scale', 'shift', 'mean', 'invstd')}
    gd, bd = gam[0].to(U.dev()), bet[0].to(U.dev())
    rm, rv = torch.zeros(Cc, device=U.dev()), torch.ones(Cc, device=U.dev())
    nbt = torch.zeros((), dtype=torch.long, device=U.dev())
    p = L.RdBnFwd()
    p.stats = stats.data_ptr()
    for k, t in bufs.items():
        setattr(p, k, t.data_ptr())
    for g in range(G):                                     # both groups share one BN, like the two encoder passes
        p.gamma[g], p.beta[g], p.running_mean[g], p.running_var[g] = gd.data_ptr(), bd.data_ptr(), rm.data_ptr(), rv.data_ptr()
        p.num_batches_tracked[g] = nbt.data_ptr()
        p.count[g] = (gstart[g + 1] - gstart[g]) * H * W
    p.C, p.G, p.eps, p.momentum, p.training = Cc, G, 1e-5, 0.1, 1
    L.check(L.lib().rd_bn_finalize_fwd(C.byref(p), None), 'bnf')
    torch.cuda.synchronize()
    bn = torch.nn.BatchNorm2d(Cc)
    bn.weight.data, bn.bias.data = gam[0].clone(), bet[0].clone()
    bn.train()
    for g in range(G):
        xs = x[gstart[g]:gstart[g + 1]]
        ref = bn(xs)
        got = xs * bufs['scale'][g].cpu()[None, :, None, None] + bufs['shift'][g].cpu()[None, :, None, None]
        np.testing.assert_allclose(got.detach(), ref.detach(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rm.cpu(), bn.running_mean, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv.cpu(), bn.running_var, rtol=1e-5, atol=1e-6)
    assert int(nbt) == 2
    # eval mode: running statistics
    p.training = 0
    L.check(L.lib().rd_bn_finalize_fwd(C.byref(p), None), 'bnf-eval')
    torch.cuda.synchronize()
    bn.eval()
    got = x * bufs['scale'][0].cpu()[None, :, None, None] + bufs['shift'][0].cpu()[None, :, None, None]
    np.testing.assert_allclose(got, bn(x).detach(), rtol=1e-4, atol=1e-5)
    assert int(nbt) == 2
    # backward coefficients: dz = P*g + Q*z + R against autograd through F.batch_norm, group by group
    p.training = 1
    rm.zero_(); rv.fill_(1)
    L.check(L.lib().rd_bn_finalize_fwd(C.byref(p), None), 'bnf2')
    gy = torch.randn(N, Cc, H, W, generator=gen)
    bst = torch.stack([torch.stack([gy[gstart[g]:gstart[g + 1]].sum((0, 2, 3)), (gy * x)[gstart[g]:gstart[g + 1]].sum((0, 2, 3))], -1)
                       for g in range(G)])
    bst = _slotted(bst).to(U.dev())
    q = L.RdBnBwd()
    dgam, dbet = torch.zeros(Cc, device=U.dev()), torch.zeros(Cc, device=U.dev())
    PQR = [torch.zeros(G, Cc, device=U.dev()) for _ in range(3)]
    q.bstats, q.mean, q.invstd = bst.data_ptr(), bufs['mean'].data_ptr(), bufs['invstd'].data_ptr()
    q.P, q.Q, q.R = (t.data_ptr() for t in PQR)
    for g in range(G):
        q.gamma[g], q.dgamma[g], q.dbeta[g] = gd.data_ptr(), dgam.data_ptr(), dbet.data_ptr()
        q.count[g] = (gstart[g + 1] - gstart[g]) * H * W
    q.C, q.G = Cc, G
    L.check(L.lib().rd_bn_finalize_bwd(C.byref(q), None), 'bnb')
    torch.cuda.synchronize()
    gw = gam[0].clone().requires_grad_(True)
    gb = bet[0].clone().requires_grad_(True)
    for g in range(G):
        xs = x[gstart[g]:gstart[g + 1]].clone().requires_grad_(True)
        (F.batch_norm(xs, None, None, gw, gb, True) * gy[gstart[g]:gstart[g + 1]]).sum().backward()
        P, Q, R = (t[g].cpu()[None, :, None, None] for t in PQR)
        got = P * gy[gstart[g]:gstart[g + 1]] + Q * xs.detach() + R
        np.testing.assert_allclose(got, xs.grad, rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(dgam.cpu(), gw.grad, rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(dbet.cpu(), gb.grad, rtol=1e-4, atol=1e-4)


def _bn_standalone(x_nchw, gstart, dtype='f32'):
    """rd_bn_stats + rd_bn_finalize_fwd on a plain tensor -> (mean, invstd) [G][C] (what a standalone BatchNorm2d does)."""
    N, Cc, H, W = x_nchw.shape
    G = len(gstart) - 1
    xd = U.nhwc(x_nchw, dtype)
    stats = torch.zeros(G, L.STAT_SLOTS, Cc, 2, dtype=torch.float64, device=U.dev())
    L.check(L.lib().rd_bn_stats(L.ptr(xd), L.ptr(stats), N, H, W, Cc, G, L.gstart_array(gstart), U.DT[dtype][0], None), 'bn_stats')
    bufs = {k: torch.zeros(G, Cc, device=U.dev()) for k in ('scale', 'shift', 'mean', 'invstd')}
    one, zero = torch.ones(Cc, device=U.dev()), torch.zeros(Cc, device=U.dev())
    p = L.RdBnFwd()
    p.stats = stats.data_ptr()
    for k, t in bufs.items():
        setattr(p, k, t.data_ptr())
    for g in range(G):
        p.gamma[g], p.beta[g] = one.data_ptr(), zero.data_ptr()
        p.count[g] = (gstart[g + 1] - gstart[g]) * H * W
    p.C, p.G, p.eps, p.momentum, p.training = Cc, G, 1e-5, 0.1, 1
    L.check(L.lib().rd_bn_finalize_fwd(C.byref(p), None), 'bnf')
    torch.cuda.synchronize()
    return bufs['mean'].cpu(), bufs['invstd'].cpu(), stats


def test_bn_stats_standalone_and_large_mean_over_sigma():
    """ATen's batch_norm never forms E[x^2]-E[x]^2 in fp32 (mean first, then deviations / Welford).  A channel with
    |mean| = 1e3 sigma (and one with 3e4 sigma) must still get the right variance: rd_bn_stats sums deviations from a
    per-thread pivot and everything across threads is fp64."""
    gen = torch.Generator().manual_seed(11)
    N, Cc, H, W = 4, 16, 40, 56
    gstart = [0, 1, 4]
    x = torch.randn(N, Cc, H, W, generator=gen)
    x[:, 3] = x[:, 3] * 1e-3 + 1.0                    # mean / sigma = 1e3
    x[:, 5] = x[:, 5] + 3e4                           # mean / sigma = 3e4 (fp32 resolution of x itself: 2e-3)
    x[:, 7] = 2.5                                     # constant channel: variance exactly 0
    mean, invstd, _ = _bn_standalone(x, gstart)
    for g in range(2):
        xs = x[gstart[g]:gstart[g + 1]].double()
        m = xs.mean((0, 2, 3))
        v = xs.var((0, 2, 3), unbiased=False)
        sd = torch.sqrt(v)
        assert bool(((mean[g].double() - m).abs() <= 1e-6 * torch.maximum(m.abs(), sd)).all())   # fp32 output: 1e-6 of max(|mean|, sigma)
        np.testing.assert_allclose(invstd[g], 1 / torch.sqrt(v + 1e-5), rtol=2e-5)


@pytest.mark.parametrize('dtype', DTYPES)
def test_conv_bn_statistics_with_a_large_bias(dtype):
    """A conv bias 1e3 x the spread of the conv result (the one structural way a BatchNorm input gets |mean| >> sigma):
    the epilogue's sums exclude the bias and rd_bn_finalize_fwd adds it back -> mean and invstd as torch computes them."""
    gen = torch.Generator().manual_seed(12)
    keep = U.Keep()
    N, H, W, Cin, Cout, gstart = 2, 24, 40, 16, 32, [0, 1, 2]
    a = U.rnd(torch.randn(N, Cin, H, W, generator=gen), dtype)
    w = U.rnd(torch.randn(Cout, Cin, 3, 3, generator=gen) / np.sqrt(Cin * 9), dtype)
    bias = 1e3 * (1 + torch.rand(Cout, generator=gen))
    srcs = [U.make_src(keep, a, L.SRC_RAW, dtype)]
    p = _conv_desc(keep, srcs, w, bias, N, H, W, gstart, dtype, 9)
    out = torch.empty((N, H, W, Cout), dtype=U.DT[dtype][1], device=U.dev())
    stats = torch.zeros(2, L.STAT_SLOTS, Cout, 2, dtype=torch.float64, device=U.dev())
    p.emode, p.out, p.stats = 0, out.data_ptr(), stats.data_ptr()
    L.check(L.lib().rd_conv(C.byref(p), U.DT[dtype][0], None), 'conv')
    bufs = {k: torch.zeros(2, Cout, device=U.dev()) for k in ('scale', 'shift', 'mean', 'invstd')}
    one, zero, bd = torch.ones(Cout, device=U.dev()), torch.zeros(Cout, device=U.dev()), bias.to(U.dev())
    q = L.RdBnFwd()
    q.stats, q.conv_bias = stats.data_ptr(), bd.data_ptr()
    for k, t in bufs.items():
        setattr(q, k, t.data_ptr())
    for g in range(2):
        q.gamma[g], q.beta[g] = one.data_ptr(), zero.data_ptr()
        q.count[g] = H * W
    q.C, q.G, q.eps, q.momentum, q.training = Cout, 2, 1e-5, 0.1, 1
    L.check(L.lib().rd_bn_finalize_fwd(C.byref(q), None), 'bnf')
    torch.cuda.synchronize()
    ref = F.conv2d(a.double(), w.double(), None, padding=1)
    for g in range(2):
        m = ref[g:g + 1].mean((0, 2, 3)) + bias.double()
        v = ref[g:g + 1].var((0, 2, 3), unbiased=False)
        np.testing.assert_allclose(bufs['mean'][g].cpu(), m, rtol=1e-6)
        np.testing.assert_allclose(bufs['invstd'][g].cpu(), 1 / torch.sqrt(v + 1e-5), rtol=1e-4)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('with_bn', [True, False])
def test_pool_materialised_forward_and_backward(dtype, with_bn):
    """rd_pool_fwd / rd_pool_bwd == F.max_pool2d(act(bn(z)), 2) and its autograd backward (nn.MaxPool2d(2), unet.py:45,56):
    first maximum on ties (a ReLU'd window of non-positive values is a 4-way tie at 0), activation mask, accumulation onto an
    existing gradient, BatchNorm-backward sums of the scattered values."""
    gen = torch.Generator().manual_seed(21)
    N, Cc, Ho, Wo = 3, 16, 7, 9
    gstart, slope = [0, 1, 3], 0.0
    z = U.rnd(torch.randn(N, Cc, 2 * Ho, 2 * Wo, generator=gen), dtype)
    z[0, :, 0:2, 0:2] = -1.0                                            # all four activations 0: tie -> position (0, 0)
    z[1, 3, 2:4, 2:4] = 0.75                                            # exact tie of positive values
    sc, sh = _params(2, Cc, gen)
    zd = U.nhwc(z, dtype)
    scd, shd = (U.fdev(sc), U.fdev(sh)) if with_bn else (None, None)
    gs = L.gstart_array(gstart)
    out = torch.full((N, Ho, Wo, Cc), float('nan'), dtype=U.DT[dtype][1], device=U.dev())
    L.check(L.lib().rd_pool_fwd(L.ptr(zd), L.ptr(scd), L.ptr(shd), slope if with_bn else 1.0, L.ptr(out), N, Ho, Wo, Cc, 2, gs,
                                U.DT[dtype][0], None, 0, None), 'pool_fwd')
    torch.cuda.synchronize()
    zz = z.clone().requires_grad_(True)
    pre = (zz * U.group_rows(sc, gstart, N) + U.group_rows(sh, gstart, N)) if with_bn else zz
    a = F.relu(pre) if with_bn else pre
    ref = F.max_pool2d(a, 2)
    U.assert_close(U.from_nhwc(out), ref.detach(), dtype, 'pool_fwd')
    gp = U.rnd(torch.randn(N, Cc, Ho, Wo, generator=gen), dtype)
    old = U.rnd(torch.randn(N, Cc, 2 * Ho, 2 * Wo, generator=gen), dtype)
    pre.retain_grad()
    ref.backward(gp)
    gref = pre.grad if with_bn else zz.grad                             # gradient w.r.t. the BN output (what g holds)
    for accumulate in (0, 1):
        gbuf = U.nhwc(old, dtype)
        bst = torch.zeros(2, L.STAT_SLOTS, Cc, 2, dtype=torch.float64, device=U.dev())
        L.check(L.lib().rd_pool_bwd(L.ptr(U.nhwc(gp, dtype)), L.ptr(zd), L.ptr(scd), L.ptr(shd), slope if with_bn else 1.0, 1 if with_bn else 0,
                                    L.ptr(gbuf), accumulate, L.ptr(bst) if with_bn else None, N, Ho, Wo, Cc, 2, gs, U.DT[dtype][0],
                                    accumulate * L.STAT_SLOTS_FOLD, None), 'pool_bwd')
        torch.cuda.synchronize()
        U.assert_close(U.from_nhwc(gbuf), gref + (old if accumulate else 0), dtype, 'pool_bwd acc=%d' % accumulate)
        if with_bn:
            rs = torch.stack([torch.stack([gref[gstart[g]:gstart[g + 1]].sum((0, 2, 3)), (gref * z)[gstart[g]:gstart[g + 1]].sum((0, 2, 3))], -1)
                              for g in range(2)])
            np.testing.assert_allclose(bst.sum(1).cpu().float(), rs, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize('dtype', DTYPES)
def test_upsample_stats_and_backward(dtype):
    gen = torch.Generator().manual_seed(5)
    N, Cc, h, w = 3, 32, 7, 9
    gstart = [0, 1, 3]
    t = U.rnd(torch.randn(N, Cc, h, w, generator=gen), dtype).requires_grad_(True)
    y = F.interpolate(t, scale_factor=2, mode='bilinear', align_corners=False)
    td = U.nhwc(t.detach(), dtype)
    stats = torch.zeros(2, L.STAT_SLOTS, Cc, 2, dtype=torch.float64, device=U.dev())
    gs = L.gstart_array(gstart)
    L.check(L.lib().rd_up_stats(L.ptr(td), L.ptr(stats), None, N, h, w, Cc, 2, gs, U.DT[dtype][0], 0, None), 'upstats')
    torch.cuda.synchronize()
    ref = torch.stack([torch.stack([y[gstart[g]:gstart[g + 1]].sum((0, 2, 3)), y[gstart[g]:gstart[g + 1]].pow(2).sum((0, 2, 3))], -1)
                       for g in range(2)]).detach()
    np.testing.assert_allclose(stats.sum(1).cpu().float(), ref, rtol=1e-3, atol=1e-2)
    # materialising variant: y is stored, and the statistics are those of the stored (rounded) values
    stats2 = torch.zeros_like(stats)
    yd = torch.full((N, 2 * h, 2 * w, Cc), float('nan'), dtype=U.DT[dtype][1], device=U.dev())
    L.check(L.lib().rd_up_stats(L.ptr(td), L.ptr(stats2), L.ptr(yd), N, h, w, Cc, 2, gs, U.DT[dtype][0], L.STAT_SLOTS_FOLD, None), 'upstats+y')
    torch.cuda.synchronize()
    assert float(stats2[:, L.STAT_SLOTS_FOLD:].abs().max()) == 0.0            # only the first RD_STAT_SLOTS_FOLD copies were used
    U.assert_close(U.from_nhwc(yd), y.detach(), dtype, 'up y')
    ys = U.from_nhwc(yd)
    ref2 = torch.stack([torch.stack([ys[gstart[g]:gstart[g + 1]].sum((0, 2, 3)), ys[gstart[g]:gstart[g + 1]].pow(2).sum((0, 2, 3))], -1)
                        for g in range(2)])
    np.testing.assert_allclose(stats2.sum(1).cpu().float(), ref2, rtol=1e-4, atol=1e-3)
    P, R = _params(2, Cc, gen)
    Q = 0.1 * torch.randn(2, Cc, generator=gen)
    g2 = U.rnd(torch.randn(N, Cc, 2 * h, 2 * w, generator=gen), dtype)
    dzh = g2 * U.group_rows(P, gstart, N) + y.detach() * U.group_rows(Q, gstart, N) + U.group_rows(R, gstart, N)
    (y * dzh).sum().backward()
    dt = torch.full((N, h, w, Cc), float('nan'), dtype=U.DT[dtype][1], device=U.dev())
    g2d = U.nhwc(g2, dtype)
    Pd, Qd, Rd = U.fdev(P), U.fdev(Q), U.fdev(R)
    L.check(L.lib().rd_up_bwd(L.ptr(g2d), L.ptr(td), L.ptr(dt), L.ptr(Pd), L.ptr(Qd), L.ptr(Rd), N, h, w, Cc, 2, gs,
                              U.DT[dtype][0], None, 0, None), 'upbwd')
    torch.cuda.synchronize()
    U.assert_close(U.from_nhwc(dt), t.grad, dtype, 'up_bwd')


# ------------------------------------------------------------------------------------ losses / Adam / layout
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('cons', [0, 1, 2])
def test_seg_loss_fundus(dtype, cons):
    from oracle import losses as OL
    gen = torch.Generator().manual_seed(21 + cons)
    B, K, H, W = 3, 2, 12, 20
    lg = U.rnd(2 * torch.randn(2 * B, K, H, W, generator=gen), dtype).requires_grad_(True)
    mask = (torch.rand(B, K, H, W, generator=gen) > 0.6).float()
    p1, p2 = torch.sigmoid(lg[:B]), torch.sigmoid(lg[B:])
    comps = [OL.bce(p1, mask), OL.dice_loss(p1, mask), OL.bce(p2, mask), OL.dice_loss(p2, mask)]
    c = OL.kd(p2, p1) if cons == 1 else (F.mse_loss(p2, p1) if cons == 2 else torch.zeros(()))
    total = sum(comps) + 0.5 * c
    total.backward()
    _run_seg(lg, mask.to(U.dev()), B, K, H, W, 0, cons, dtype, comps + [c, total])


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('cons', [1, 2])
def test_seg_loss_prostate(dtype, cons):
    from oracle import losses as OL
    gen = torch.Generator().manual_seed(31 + cons)
    B, K, H, W = 2, 2, 10, 18
    lg = U.rnd(2 * torch.randn(2 * B, K, H, W, generator=gen), dtype).requires_grad_(True)
    tgt = (torch.rand(B, H, W, generator=gen) > 0.7).long()
    p1, p2 = torch.softmax(lg[:B], 1), torch.softmax(lg[B:], 1)
    comps = [F.cross_entropy(lg[:B], tgt), OL.dice_loss_multi(p1, tgt, K, 0), F.cross_entropy(lg[B:], tgt), OL.dice_loss_multi(p2, tgt, K, 0)]
    c = OL.kd(p2, p1) if cons == 1 else F.mse_loss(p2, p1)
    total = sum(comps) + 0.5 * c
    total.backward()
    _run_seg(lg, tgt.to(U.dev()), B, K, H, W, 1, cons, dtype, comps + [c, total])


def _run_seg(lg, target_dev, B, K, H, W, kind, cons, dtype, ref_losses):
    for cs in (K, 8):                        # dense dlogits, and dlogits padded to one 16-byte slot
        _run_seg_stride(lg, target_dev, B, K, H, W, kind, cons, dtype, ref_losses, cs)


def _run_seg_stride(lg, target_dev, B, K, H, W, kind, cons, dtype, ref_losses, cs):
    lgd = U.nhwc(lg.detach(), dtype)
    dl = torch.full((2 * B, H, W, cs), float('nan'), dtype=U.DT[dtype][1], device=U.dev())
    losses = torch.zeros(8, device=U.dev())
    p = L.RdSegLoss()
    p.logits, p.target, p.dlogits, p.losses_out = lgd.data_ptr(), target_dev.data_ptr(), dl.data_ptr(), losses.data_ptr()
    p.B, p.H, p.W, p.K, p.kind, p.consistency, p.cons_weight = B, H, W, K, kind, cons, 0.5
    p.dlogits_cstride = 0 if cs == K else cs
    ws = torch.empty(L.lib().rd_seg_loss_workspace(C.byref(p)) // 4, device=U.dev())
    p.partial = ws.data_ptr()
    L.check(L.lib().rd_seg_loss(C.byref(p), U.DT[dtype][0], None), 'segloss')
    torch.cuda.synchronize()
    np.testing.assert_allclose(losses[:6].cpu(), [float(v) for v in ref_losses], rtol=2e-5, atol=1e-7)
    U.assert_close(U.from_nhwc(dl[..., :K]), lg.grad, dtype, 'dlogits', scale=0.5 if dtype == 'bf16' else 5.0)
    assert bool(torch.isnan(dl[..., K:].float()).all())          # the pad is never written


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('cs', [3, 8])
def test_rec_loss(dtype, cs):
    gen = torch.Generator().manual_seed(41)
    B, Cc, H, W = 5, 3, 8, 12
    gstart = [0, 2, 3, 5]
    lg = U.rnd(torch.randn(B, Cc, H, W, generator=gen), dtype).requires_grad_(True)
    tgt = U.rnd(torch.rand(B, Cc, H, W, generator=gen) * 2 - 1, dtype)
    mses = [F.mse_loss(torch.tanh(lg[gstart[g]:gstart[g + 1]]), tgt[gstart[g]:gstart[g + 1]]) for g in range(3)]
    (0.1 * sum(mses)).backward()
    lgd, td = U.nhwc(lg.detach(), dtype), U.nhwc(tgt, dtype)
    if cs != Cc:                              # padded target (the network input) and padded dlogits
        td = torch.cat([td, torch.full((B, H, W, cs - Cc), 7.0, dtype=td.dtype, device=td.device)], -1).contiguous()
    dl = torch.full((B, H, W, cs), float('nan'), dtype=lgd.dtype, device=lgd.device)
    out = torch.zeros(3, device=U.dev())
    ws = torch.empty(L.lib().rd_rec_loss_workspace(B, H, W, Cc) // 4, device=U.dev())
    L.check(L.lib().rd_rec_loss(L.ptr(lgd), L.ptr(td), L.ptr(dl), L.ptr(out), L.ptr(ws), B, H, W, Cc, cs, cs, 3, L.gstart_array(gstart), 0.1,
                                U.DT[dtype][0], None), 'recloss')
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu(), [float(m) for m in mses], rtol=2e-5)
    U.assert_close(U.from_nhwc(dl[..., :Cc]), lg.grad, dtype, 'rec dlogits', scale=0.5 if dtype == 'bf16' else 5.0)


def test_adam_matches_torch_optim_with_poly_lr():
    gen = torch.Generator().manual_seed(51)
    n, n_enc, base_lr, total = 1000, 300, 2e-3, 40
    p0 = torch.randn(n, generator=gen)
    pe = torch.nn.Parameter(p0[:n_enc].clone())
    po = torch.nn.Parameter(p0[n_enc:].clone())
    opt = torch.optim.Adam([{'params': [pe], 'lr': base_lr / 2}, {'params': [po], 'lr': base_lr}], lr=base_lr, betas=(0.9, 0.999))
    pd, m, v = p0.clone().to(U.dev()), torch.zeros(n, device=U.dev()), torch.zeros(n, device=U.dev())
    it = torch.zeros((), dtype=torch.int32, device=U.dev())
    hyper = torch.zeros(4, device=U.dev())
    a = L.RdAdam()
    a.param, a.exp_avg, a.exp_avg_sq, a.n, a.n_half_lr = pd.data_ptr(), m.data_ptr(), v.data_ptr(), n, n_enc
    a.iter, a.hyper_out, a.base_lr, a.total_iters, a.beta1, a.beta2, a.eps = it.data_ptr(), hyper.data_ptr(), base_lr, total, 0.9, 0.999, 1e-8
    for step in range(5):
        g = torch.randn(n, generator=gen)
        pe.grad, po.grad = g[:n_enc].clone(), g[n_enc:].clone()
        opt.step()
        lr = base_lr * (1 - step / total) ** 0.9           # train.py:289-293, written after the step
        opt.param_groups[0]['lr'], opt.param_groups[1]['lr'] = lr / 2, lr
        gd = g.to(U.dev())
        a.grad = gd.data_ptr()
        L.check(L.lib().rd_adam_step(C.byref(a), None), 'adam')
        torch.cuda.synchronize()
        ref = torch.cat([pe.data, po.data])
        np.testing.assert_allclose(pd.cpu(), ref, rtol=1e-5, atol=2e-7)
    assert int(it) == 5


@pytest.mark.parametrize('dtype', DTYPES)
def test_layout_boundary_kernels(dtype):
    gen = torch.Generator().manual_seed(61)
    N, Cc, H, W = 3, 16, 9, 11
    gstart = [0, 1, 3]
    x = U.rnd(torch.randn(N, Cc, H, W, generator=gen), dtype)
    xd = x.to(U.dev())
    y = torch.empty(N, H, W, Cc, dtype=U.DT[dtype][1], device=U.dev())
    L.check(L.lib().rd_nchw_to_nhwc(L.ptr(xd), L.ptr(y), N, Cc, H, W, 0, U.DT[dtype][0], None), 'to_nhwc')
    torch.cuda.synchronize()
    assert torch.equal(U.from_nhwc(y), x)
    yp = torch.zeros(N, H, W, 8, dtype=U.DT[dtype][1], device=U.dev())        # 3 channels into a padded slot
    L.check(L.lib().rd_nchw_to_nhwc(L.ptr(xd[:, :3].contiguous()), L.ptr(yp), N, 3, H, W, 8, U.DT[dtype][0], None), 'to_nhwc pad')
    torch.cuda.synchronize()
    assert torch.equal(U.from_nhwc(yp[..., :3]), x[:, :3]) and float(yp[..., 3:].float().abs().max()) == 0.0
    sc, sh = _params(2, Cc, gen)
    scd, shd = U.fdev(sc), U.fdev(sh)
    back = torch.empty(N, Cc, H, W, device=U.dev())
    gs = L.gstart_array(gstart)
    L.check(L.lib().rd_nhwc_to_nchw(L.ptr(y), L.ptr(back), L.ptr(scd), L.ptr(shd), 1, 0.0, N, Cc, H, W, 2, gs, U.DT[dtype][0], None), 'to_nchw')
    torch.cuda.synchronize()
    ref = F.relu(x * U.group_rows(sc, gstart, N) + U.group_rows(sh, gstart, N))
    np.testing.assert_allclose(back.cpu(), ref, rtol=1e-6, atol=1e-6)
    # gradient entering from torch: mask + BN-backward sums
    dy = torch.randn(N, Cc, H, W, generator=gen)
    old = U.rnd(torch.randn(N, Cc, H, W, generator=gen), dtype)
    gbuf = U.nhwc(old, dtype)
    bst = torch.zeros(2, L.STAT_SLOTS, Cc, 2, dtype=torch.float64, device=U.dev())
    dyd = dy.to(U.dev())
    L.check(L.lib().rd_grad_in(L.ptr(dyd), L.ptr(y), L.ptr(gbuf), L.ptr(scd), L.ptr(shd), L.ptr(bst), 1, 0.0, 1, N, Cc, H, W, 2, gs,
                               U.DT[dtype][0], None), 'grad_in')
    torch.cuda.synchronize()
    gnew = dy * (ref > 0).float()
    U.assert_close(U.from_nhwc(gbuf), old + gnew, dtype, 'grad_in')
    rs = torch.stack([torch.stack([gnew[gstart[g]:gstart[g + 1]].sum((0, 2, 3)), (gnew * x)[gstart[g]:gstart[g + 1]].sum((0, 2, 3))], -1)
                      for g in range(2)])
    np.testing.assert_allclose(bst.sum(1).cpu().float(), rs, rtol=1e-3, atol=1e-3)
    # column sums (bias gradient of out1)
    t3 = U.rnd(torch.randn(2, 3, 20, 30, generator=gen), dtype)
    t3d = U.nhwc(t3, dtype)
    outc = torch.ones(3, device=U.dev())
    wsb = torch.empty(8192, device=U.dev())
    L.check(L.lib().rd_colsum(L.ptr(t3d), L.ptr(outc), L.ptr(wsb), 2 * 20 * 30, 3, 0, 1.0, U.DT[dtype][0], None), 'colsum')
    torch.cuda.synchronize()
    np.testing.assert_allclose(outc.cpu(), 1 + t3.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    t3p = torch.cat([t3d, torch.ones(2, 20, 30, 5, dtype=t3d.dtype, device=t3d.device)], -1).contiguous()
    L.check(L.lib().rd_colsum(L.ptr(t3p), L.ptr(outc), L.ptr(wsb), 2 * 20 * 30, 3, 8, 0.0, U.DT[dtype][0], None), 'colsum pad')
    torch.cuda.synchronize()
    np.testing.assert_allclose(outc.cpu(), t3.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)


# ------------------------------------------------------------------------------------ every 64-wide conv kernel
@pytest.mark.parametrize('env', [
    {'RD_CONV_WS': '3', 'RD_CONV_WS_MIN2': '0', 'RD_CONV_NB1_BELOW': '0'},          # conv_ws_kernel: forward AND gradient launches
    {'RD_CONV_WS': '3', 'RD_CONV_WS_MIN2': '0', 'RD_CONV_NB1_BELOW': '0', 'RD_CONV_WS_FLAT': '0'},   # ... on 8 x 32 tiles only
    {'RD_CONV_WS': '3', 'RD_CONV_WS_MIN2': '1073741824', 'RD_CONV_NB1_BELOW': '0'},  # conv_ws_kernel forward, gradients on conv_pf_kernel (the round-3 default)
    {'RD_CONV_WS': '0', 'RD_CONV_PP_ALL': '1', 'RD_CONV_NB1_BELOW': '0'},           # conv_pp_kernel (LDS-staged epilogue) everywhere
    {'RD_CONV_PP_OFF': '1', 'RD_CONV_NB1_BELOW': '0'},                              # conv_pf_kernel with the register epilogues
    {'RD_CONV_PP_OFF': '1', 'RD_CONV_NB1_BELOW': '0', 'RD_CONV_FLAT_TILES': '0'},   # ... on 8 x 32 tiles only (default: 10 x 25 where more lanes are live)
    {'RD_CONV_PP_OFF': '1', 'RD_CONV_PF_LEAN_OFF': '1', 'RD_CONV_NB1_BELOW': '0'},  # conv_pf_kernel with the LDS-staged epilogue
    {'RD_CONV_NB1_BELOW': '100000'},                                                # 32-wide tiles for every 64-wide launch
    {'RD_SW_TPW': '5'},                                                             # conv_small_fwd_kernel: 5 tiles per workgroup (+ ghosts)
    {'RD_SW_TPW': '2'},                                                             # ... fewer tiles than register sets
    {'RD_SW_NWV': '4'},                                                             # ... two 256-thread workgroups per CU, two tile rows per wave (<= 16-channel inputs)
    {'RD_SW_NWV': '4', 'RD_SW_TPW': '5'},
    {'RD_SW_NWV': '8'},                                                             # ... one 512-thread workgroup per CU for every input width
    {'RD_CONV_SMALL_FWD': '0'},                                                      # conv_small_kernel for the forward launches
], ids=['ws_fwd_bwd', 'ws_fwd_bwd_8x32', 'ws_fwd_pf_bwd', 'pp_staged', 'pf_lean', 'pf_lean_8x32', 'pf_staged', 'nb1_everywhere', 'small_fwd_tpw5', 'small_fwd_tpw2', 'small_fwd_nwv4', 'small_fwd_nwv4_tpw5', 'small_fwd_nwv8', 'small_fwd_off'])
def test_conv_kernels_under_forced_dispatch(env):
    """Which kernel a 64-wide launch takes depends on its size (csrc/conv_pp.hip, conv_big.hip), and the cases above are small.
    The dispatch switches (debug build of the library only) are read once per process: re-run the conv parity tests in a child
    process with every eligible launch forced through each kernel / epilogue / tile width."""
    import subprocess
    import sys
    e = dict(os.environ, RAMDSIR_DEBUG_LIB='1')        # the RD_* overrides exist only in the debug build (csrc/common.h rd_switch)
    e.update(env)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-m', 'gpu', '-k',
                        'test_conv_forward or test_conv_gradient_epilogues or test_conv_bnbwd'], env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    tail = r.stdout.decode()[-2500:]
    assert r.returncode == 0, tail
    assert ' passed' in tail and 'failed' not in tail, tail


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['three_aligned', 'ragged_sizes', 'nine_ranges', 'unaligned_start'])
def test_rd_zero_clears_exactly_its_ranges(case):
    """rd_zero (include/ramdsir.h; optimizer.zero_grad() of code/train.py:285,454 + the per-step reset of the BatchNorm sum arenas): up to
    8 ranges with 16-byte-aligned starts in one kernel launch, anything else through the runtime's fills -- every byte of every range
    zero, every byte outside untouched."""
    buf = torch.full((1 << 20,), 0x5a, dtype=torch.uint8, device=U.dev())
    spans = dict(three_aligned=[(0, 4096), (65536, 300000), (524288, 16)],
                 ragged_sizes=[(16, 1), (1024, 17), (4096, 100003), (262144, 15)],
                 nine_ranges=[(4096 * i, 1000 + i) for i in range(9)],
                 unaligned_start=[(3, 1000), (8192, 4096)])[case]
    base = buf.data_ptr()
    assert base % 16 == 0
    ptrs = (L.vp * len(spans))(*[base + o for o, _ in spans])
    sizes = (L.i64 * len(spans))(*[n for _, n in spans])
    L.check(L.lib().rd_zero(ptrs, sizes, len(spans), None), case)
    torch.cuda.synchronize()
    want = torch.full((1 << 20,), 0x5a, dtype=torch.uint8)
    for o, n in spans:
        want[o:o + n] = 0
    assert torch.equal(buf.cpu(), want)
