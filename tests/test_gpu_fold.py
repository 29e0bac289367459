"""-m gpu: the BatchNorm finalize FOLDED into its consumer (rd_src_t.fin, csrc/bn_fin.h, include/ramdsir.h) against the explicit
rd_bn_finalize_fwd / rd_bn_finalize_bwd launches it replaces -- nn.BatchNorm2d in training mode as the reference uses it
(code/networks/unet.py:17-28; DomainSpecificBatchNorm2d code/networks/dsbn.py:10-11,24-27): bit for bit, on every kernel family a
folded launch can be routed to, for groups that share one BatchNorm (the two passes of the seg network: running statistics updated
twice, in group order) and for one BatchNorm per group (DSBN)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from ramdsir import _lib as L          # noqa: E402
import gpu_util as U                   # noqa: E402
from test_gpu_ops import _conv_desc, _params   # noqa: E402


def _dev_copy(keep, desc):
    """rd_src_t.fin: the address of the (host) finalize descriptor; the entry point copies it into the kernel arguments."""
    keep(desc)
    return C.addressof(desc)


def _bn_state(keep, G, Cc, shared, gen):
    """Parameters and buffers of the BatchNorm(s) behind G groups: one shared module, or one per group (DSBN)."""
    nmod = 1 if shared else G
    st = dict(gamma=[keep(U.fdev(1.0 + 0.2 * torch.randn(Cc, generator=gen))) for _ in range(nmod)],
              beta=[keep(U.fdev(0.3 * torch.randn(Cc, generator=gen))) for _ in range(nmod)],
              rm=[keep(U.fdev(0.1 * torch.randn(Cc, generator=gen))) for _ in range(nmod)],
              rv=[keep(U.fdev(1.0 + 0.1 * torch.rand(Cc, generator=gen))) for _ in range(nmod)],
              nbt=[keep(torch.full((), 3, dtype=torch.long, device=U.dev())) for _ in range(nmod)],
              dgamma=[keep(U.fdev(0.05 * torch.randn(Cc, generator=gen))) for _ in range(nmod)],
              dbeta=[keep(U.fdev(0.05 * torch.randn(Cc, generator=gen))) for _ in range(nmod)])
    st['mod'] = [0] * G if shared else list(range(G))
    return st


def _random_slots(G, Cc, nslots, gen, scale=100.0):
    """Statistic buffers as producers leave them: fp64 sums of fp32 terms in the first `nslots` copies, the rest zero."""
    st = torch.zeros(G, L.STAT_SLOTS, Cc, 2, dtype=torch.float64)
    s1 = (torch.randn(G, nslots, Cc, generator=gen) * scale).float().double()
    s2 = (torch.rand(G, nslots, Cc, generator=gen) * scale * 40 + scale * 30).float().double()      # sum of squares: dominates mean^2
    st[:, :nslots, :, 0], st[:, :nslots, :, 1] = s1, s2
    return st.to(U.dev())


def _fwd_desc(keep, stats, st, Cc, counts, bias, nslots):
    G = len(counts)
    bufs = {k: keep(torch.full((G, Cc), float('nan'), device=U.dev())) for k in ('scale', 'shift', 'mean', 'invstd')}
    p = L.RdBnFwd()
    p.stats = stats.data_ptr()
    p.conv_bias = bias.data_ptr() if bias is not None else None
    for k, t in bufs.items():
        setattr(p, k, t.data_ptr())
    for g in range(G):
        m = st['mod'][g]
        p.gamma[g], p.beta[g] = st['gamma'][m].data_ptr(), st['beta'][m].data_ptr()
        p.running_mean[g], p.running_var[g] = st['rm'][m].data_ptr(), st['rv'][m].data_ptr()
        p.num_batches_tracked[g] = st['nbt'][m].data_ptr()
        p.count[g] = float(counts[g])
    p.C, p.G, p.eps, p.momentum, p.training, p.nslots = Cc, G, 1e-5, 0.1, 1, nslots
    return p, bufs


def _snapshot(st):
    return {k: [t.clone() for t in st[k]] for k in ('rm', 'rv', 'nbt', 'dgamma', 'dbeta')}


def _restore(st, snap):
    for k, lst in snap.items():
        for t, s in zip(st[k], lst):
            t.copy_(s)


FWD_ROUTES = [
    # name, dtype, taps, Cin, Cout, N, H, W, gstart, shared BatchNorm
    ('small_fwd_32_32', 'bf16', 9, 32, 32, 4, 70, 100, [0, 2, 4], True),            # conv_small_fwd_kernel, several tiles per workgroup
    ('small_fwd_16_16_dsbn', 'bf16', 9, 16, 16, 3, 33, 65, [0, 1, 2, 3], False),
    ('small_k1_32_16', 'bf16', 1, 32, 16, 4, 20, 20, [0, 2, 4], True),              # conv_small_kernel (1x1)
    ('ws_64_64', 'bf16', 9, 64, 64, 8, 103, 95, [0, 3, 8], True),                   # conv_ws_kernel (persistent, >= 300 workgroups)
    ('pp_256_256', 'bf16', 9, 256, 256, 4, 25, 25, [0, 2, 4], True),                # few tiles: conv_pp_kernel / conv_pf_kernel
    ('pf_k1_128_64_dsbn', 'bf16', 1, 128, 64, 3, 10, 34, [0, 1, 2, 3], False),      # 1x1 on the chunk-pipelined kernel
    ('f32_16_16', 'f32', 9, 16, 16, 4, 16, 32, [0, 1, 4], True),                    # fp32 parity path (conv_small_kernel<float>)
    ('f32_64_32', 'f32', 9, 64, 32, 2, 9, 40, [0, 1, 2], True),                     # conv_kernel<float>
]


@pytest.mark.parametrize('case', FWD_ROUTES, ids=[c[0] for c in FWD_ROUTES])
def test_forward_finalize_folded_into_the_consuming_conv(case):
    name, dtype, taps, Cin, Cout, N, H, W, gstart, shared = case
    gen = torch.Generator().manual_seed(11)
    keep = U.Keep()
    G = len(gstart) - 1
    ns = L.STAT_SLOTS_FOLD
    x = U.rnd(torch.randn(N, Cin, H, W, generator=gen), dtype)
    st = _bn_state(keep, G, Cin, shared, gen)
    stats = _random_slots(G, Cin, ns, gen)
    cbias = keep(U.fdev(0.2 * torch.randn(Cin, generator=gen)))
    counts = [(gstart[g + 1] - gstart[g]) * H * W for g in range(G)]
    fd, bufs = _fwd_desc(keep, stats, st, Cin, counts, cbias, ns)
    w = U.rnd(torch.randn(Cout, Cin, taps == 9 and 3 or 1, taps == 9 and 3 or 1, generator=gen) / np.sqrt(Cin * taps), dtype)
    bias = 0.1 * torch.randn(Cout, generator=gen)
    src = U.make_src(keep, x, L.SRC_AFFACT, dtype, torch.zeros(G, Cin), torch.zeros(G, Cin), 0.0)
    src.scale, src.shift = bufs['scale'].data_ptr(), bufs['shift'].data_ptr()
    p = _conv_desc(keep, [src], w, bias, N, H, W, gstart, dtype, taps)
    out = torch.full((N, H, W, Cout), float('nan'), dtype=U.DT[dtype][1], device=U.dev())
    ostats = torch.zeros(G, L.STAT_SLOTS, Cout, 2, dtype=torch.float64, device=U.dev())
    p.emode, p.out, p.stats, p.stat_slots = 0, out.data_ptr(), ostats.data_ptr(), ns
    snap = _snapshot(st)
    lib = L.lib()
    # (a) explicit finalize launch, then the conv
    L.check(lib.rd_bn_finalize_fwd(C.byref(fd), None), 'finalize')
    L.check(lib.rd_conv(C.byref(p), U.DT[dtype][0], None), name)
    torch.cuda.synchronize()
    ref = dict(out=out.clone(), ostats=ostats.clone(), bufs={k: t.clone() for k, t in bufs.items()}, st=_snapshot(st))
    assert torch.isfinite(ref['out'].float()).all() and all(torch.isfinite(t).all() for t in ref['bufs'].values())
    assert float(ostats[:, ns:].abs().max()) == 0.0 and float(ostats[:, :ns].abs().max()) > 0      # rd_conv_t.stat_slots honoured
    fin = _dev_copy(keep, fd)
    for owner in (True, False):
        _restore(st, snap)
        for t in bufs.values():
            t.fill_(float('nan'))
        out.fill_(float('nan'))
        ostats.zero_()
        p.src[0].fin, p.src[0].fin_flags = fin, (L.FIN_OWNER if owner else 0)
        L.check(lib.rd_conv(C.byref(p), U.DT[dtype][0], None), name + ' folded')
        torch.cuda.synchronize()
        assert torch.equal(out, ref['out']), 'conv output'
        assert torch.equal(ostats.sum(1), ref['ostats'].sum(1))
        if owner:
            # the owner leaves the coefficient vectors in memory for the launches behind it (a non-owner may keep them to itself: kernels
            # that hand the coefficients to their loaders through LDS write nothing)
            assert torch.equal(bufs['scale'], ref['bufs']['scale']) and torch.equal(bufs['shift'], ref['bufs']['shift'])
            assert torch.equal(bufs['mean'], ref['bufs']['mean']) and torch.equal(bufs['invstd'], ref['bufs']['invstd'])
            for k in ('rm', 'rv', 'nbt'):
                for a, b in zip(st[k], ref['st'][k]):
                    assert torch.equal(a, b), k
            assert int(st['nbt'][0]) == 3 + (G if shared else 1)
        else:                                              # a non-owner leaves what happens once per BatchNorm call alone
            for k in ('rm', 'rv', 'nbt'):
                for a, b in zip(st[k], snap[k]):
                    assert torch.equal(a, b), k


@pytest.mark.parametrize('dtype', ['bf16', 'f32'])
def test_forward_finalize_folded_into_pool_and_second_source(dtype):
    """rd_pool_fwd with a fin; and a conv whose SECOND source (the concat of ConvU.conv3) carries the fin."""
    gen = torch.Generator().manual_seed(12)
    keep = U.Keep()
    N, Cc, Ho, Wo, gstart = 4, 16, 20, 24, [0, 2, 4]
    G, ns = 2, L.STAT_SLOTS_FOLD
    z = U.rnd(torch.randn(N, Cc, 2 * Ho, 2 * Wo, generator=gen), dtype)
    st = _bn_state(keep, G, Cc, True, gen)
    stats = _random_slots(G, Cc, ns, gen)
    counts = [(gstart[g + 1] - gstart[g]) * 4 * Ho * Wo for g in range(G)]
    fd, bufs = _fwd_desc(keep, stats, st, Cc, counts, None, ns)
    zd = keep(U.nhwc(z, dtype))
    gs = L.gstart_array(gstart)
    out = torch.full((N, Ho, Wo, Cc), float('nan'), dtype=U.DT[dtype][1], device=U.dev())
    lib = L.lib()
    snap = _snapshot(st)
    L.check(lib.rd_bn_finalize_fwd(C.byref(fd), None), 'finalize')
    L.check(lib.rd_pool_fwd(L.ptr(zd), L.ptr(bufs['scale']), L.ptr(bufs['shift']), 0.0, L.ptr(out), N, Ho, Wo, Cc, G, gs, U.DT[dtype][0],
                            None, 0, None), 'pool')
    torch.cuda.synchronize()
    ref = (out.clone(), {k: t.clone() for k, t in bufs.items()}, _snapshot(st))
    _restore(st, snap)
    for t in bufs.values():
        t.fill_(float('nan'))
    out.fill_(float('nan'))
    L.check(lib.rd_pool_fwd(L.ptr(zd), L.ptr(bufs['scale']), L.ptr(bufs['shift']), 0.0, L.ptr(out), N, Ho, Wo, Cc, G, gs, U.DT[dtype][0],
                            _dev_copy(keep, fd), L.FIN_OWNER, None), 'pool folded')
    torch.cuda.synchronize()
    assert torch.equal(out, ref[0])
    for k in bufs:
        assert torch.equal(bufs[k], ref[1][k]), k
    for k in ('rm', 'rv', 'nbt'):
        assert torch.equal(st[k][0], ref[2][k][0]), k
    # second source of a concat
    H, W = 2 * Ho, 2 * Wo
    skip = U.rnd(torch.randn(N, Cc, H, W, generator=gen), dtype)
    sc0, sh0 = _params(G, Cc, gen)
    s0 = U.make_src(keep, skip, L.SRC_AFFACT, dtype, sc0, sh0, 0.0)
    s1 = U.make_src(keep, z, L.SRC_AFFACT, dtype, torch.zeros(G, Cc), torch.zeros(G, Cc), 0.0)
    s1.scale, s1.shift = bufs['scale'].data_ptr(), bufs['shift'].data_ptr()
    w = U.rnd(torch.randn(32, 2 * Cc, 3, 3, generator=gen) / 17, dtype)
    p = _conv_desc(keep, [s0, s1], w, None, N, H, W, gstart, dtype, 9)
    o2 = torch.full((N, H, W, 32), float('nan'), dtype=U.DT[dtype][1], device=U.dev())
    p.emode, p.out, p.stats = 0, o2.data_ptr(), None
    L.check(lib.rd_conv(C.byref(p), U.DT[dtype][0], None), 'cat')
    torch.cuda.synchronize()
    ref2 = o2.clone()
    for t in bufs.values():
        t.fill_(float('nan'))
    o2.fill_(float('nan'))
    p.src[1].fin, p.src[1].fin_flags = _dev_copy(keep, fd), 0
    L.check(lib.rd_conv(C.byref(p), U.DT[dtype][0], None), 'cat folded')
    torch.cuda.synchronize()
    assert torch.equal(o2, ref2) and torch.isfinite(o2.float()).all()


def _bwd_desc(keep, bst, st, mean, invstd, Cc, counts, nslots):
    G = len(counts)
    PQR = [keep(torch.full((G, Cc), float('nan'), device=U.dev())) for _ in range(3)]
    q = L.RdBnBwd()
    q.bstats, q.mean, q.invstd = bst.data_ptr(), mean.data_ptr(), invstd.data_ptr()
    q.P, q.Q, q.R = (t.data_ptr() for t in PQR)
    for g in range(G):
        m = st['mod'][g]
        q.gamma[g], q.dgamma[g], q.dbeta[g] = st['gamma'][m].data_ptr(), st['dgamma'][m].data_ptr(), st['dbeta'][m].data_ptr()
        q.count[g] = float(counts[g])
    q.C, q.G, q.nslots = Cc, G, nslots
    return q, PQR


BWD_ROUTES = [
    # name, dtype, Cz (channels of dz = the conv's Cout), Ca (the conv's Cin), N, H, W, gstart, shared
    ('fused_32_32', 'bf16', 32, 32, 4, 70, 100, [0, 2, 4], True),                   # conv_small_bwd_fused_kernel / conv_small_kernel
    ('small_16_16_dsbn', 'bf16', 16, 16, 3, 33, 65, [0, 1, 2, 3], False),
    ('ws2_64_64', 'bf16', 64, 64, 7, 100, 104, [0, 2, 7], True),                    # conv_ws_kernel<2, *, true>
    ('pf_128_128', 'bf16', 128, 128, 4, 25, 25, [0, 2, 4], True),
    ('f32_16_16', 'f32', 16, 16, 4, 16, 32, [0, 1, 4], True),
]


@pytest.mark.parametrize('case', BWD_ROUTES, ids=[c[0] for c in BWD_ROUTES])
def test_backward_finalize_folded_into_gradient_and_weight_gradient_launches(case):
    name, dtype, Cz, Ca, N, H, W, gstart, shared = case
    gen = torch.Generator().manual_seed(13)
    keep = U.Keep()
    G, ns = len(gstart) - 1, L.STAT_SLOTS_FOLD
    g = U.rnd(torch.randn(N, Cz, H, W, generator=gen), dtype)
    z = U.rnd(torch.randn(N, Cz, H, W, generator=gen), dtype)
    st = _bn_state(keep, G, Cz, shared, gen)
    bst = _random_slots(G, Cz, ns, gen, scale=10.0)
    mean = keep(U.fdev(0.2 * torch.randn(G, Cz, generator=gen)))
    invstd = keep(U.fdev(0.5 + torch.rand(G, Cz, generator=gen)))
    counts = [(gstart[k + 1] - gstart[k]) * H * W for k in range(G)]
    qd, PQR = _bwd_desc(keep, bst, st, mean, invstd, Cz, counts, ns)
    w = U.rnd(torch.randn(Cz, Ca, 3, 3, generator=gen) / np.sqrt(9 * Ca), dtype)
    zprod = U.rnd(torch.randn(N, Ca, H, W, generator=gen), dtype)
    sc, sh = _params(G, Ca, gen)
    src = U.make_src(keep, g, L.SRC_BNBWD, dtype, scale=torch.zeros(G, Cz), shift=torch.zeros(G, Cz), ptr2=z, q=torch.zeros(G, Cz))
    src.scale, src.q, src.shift = PQR[0].data_ptr(), PQR[1].data_ptr(), PQR[2].data_ptr()
    p = _conv_desc(keep, [src], w, None, N, H, W, gstart, dtype, 9, transpose=True)
    p.emode, p.c_split, p.stat_slots = 1, Ca, ns
    gbuf = torch.full((N, H, W, Ca), float('nan'), dtype=U.DT[dtype][1], device=U.dev())
    dst_bst = torch.zeros(G, L.STAT_SLOTS, Ca, 2, dtype=torch.float64, device=U.dev())
    zp = keep(U.nhwc(zprod, dtype))
    d = L.RdDst()
    d.g, d.z = gbuf.data_ptr(), zp.data_ptr()
    d.scale, d.shift = keep(U.fdev(sc)).data_ptr(), keep(U.fdev(sh)).data_ptr()
    d.bstats, d.kind, d.act, d.accumulate, d.Cd, d.slope, d.n_off, d.g_fixed = dst_bst.data_ptr(), L.DST_PLAIN, 1, 0, Ca, 0.0, 0, -1
    p.dst[0] = d
    p.dst[1].kind = L.DST_NONE
    # the weight gradient of the same conv: a = act(bn(zprod)), dz = the same BatchNorm-backward pair
    wg = L.RdWgrad()
    wg.a[0] = U.make_src(keep, zprod, L.SRC_AFFACT, dtype, sc, sh, 0.0)
    wg.na, wg.taps = 1, 9
    wg.dz = src
    wg.N, wg.H, wg.W, wg.Cin, wg.Cout = N, H, W, Ca, Cz
    wg.G, wg.gstart = G, L.gstart_array(gstart)
    dW = torch.full((Cz, Ca, 3, 3), float('nan'), device=U.dev())
    lib = L.lib()
    part = keep(torch.empty(max(lib.rd_wgrad_workspace(C.byref(wg), U.DT[dtype][0]) // 4, 1) + 1, device=U.dev()))
    wg.partial, wg.dW, wg.beta = part.data_ptr(), dW.data_ptr(), 0.0
    snap = _snapshot(st)
    L.check(lib.rd_bn_finalize_bwd(C.byref(qd), None), 'finalize')
    L.check(lib.rd_conv(C.byref(p), U.DT[dtype][0], None), name)
    L.check(lib.rd_wgrad(C.byref(wg), U.DT[dtype][0], None), name + ' wgrad')
    torch.cuda.synchronize()
    ref = dict(g=gbuf.clone(), bst=dst_bst.clone(), PQR=[t.clone() for t in PQR], st=_snapshot(st), dW=dW.clone())
    assert torch.isfinite(ref['g'].float()).all() and torch.isfinite(ref['dW']).all() and all(torch.isfinite(t).all() for t in ref['PQR'])
    assert not torch.equal(ref['st']['dgamma'][0], snap['dgamma'][0])
    fin = _dev_copy(keep, qd)
    # the gradient launch as the owner
    _restore(st, snap)
    for t in PQR:
        t.fill_(float('nan'))
    gbuf.fill_(float('nan'))
    dst_bst.zero_()
    p.src[0].fin, p.src[0].fin_flags = fin, L.FIN_OWNER
    L.check(lib.rd_conv(C.byref(p), U.DT[dtype][0], None), name + ' folded')
    torch.cuda.synchronize()
    assert torch.equal(gbuf, ref['g']) and torch.equal(dst_bst.sum(1), ref['bst'].sum(1))
    for a, b in zip(PQR, ref['PQR']):
        assert torch.equal(a, b)
    for k in ('dgamma', 'dbeta'):
        for a, b in zip(st[k], ref['st'][k]):
            assert torch.equal(a, b), k
    # the weight gradient beside it: derives P / Q / R itself, leaves dgamma / dbeta alone
    _restore(st, snap)
    for t in PQR:
        t.fill_(float('nan'))
    dW.fill_(float('nan'))
    wg.dz.fin, wg.dz.fin_flags = fin, 0
    L.check(lib.rd_wgrad(C.byref(wg), U.DT[dtype][0], None), name + ' wgrad folded')
    torch.cuda.synchronize()
    assert torch.equal(dW, ref['dW'])
    for a, b in zip(PQR, ref['PQR']):                      # (a kernel that keeps the coefficients in a table of its own writes nothing:
        assert torch.equal(a, b) or bool(torch.isnan(a).all())      # wgrad_ws_kernel derives its 64 channels straight into LDS)
    for k in ('dgamma', 'dbeta'):
        for a, b in zip(st[k], snap[k]):
            assert torch.equal(a, b), k
    # ... and as the owner (the first conv of the network has no gradient launch)
    wg.dz.fin_flags = L.FIN_OWNER
    L.check(lib.rd_wgrad(C.byref(wg), U.DT[dtype][0], None), name + ' wgrad owner')
    torch.cuda.synchronize()
    for k in ('dgamma', 'dbeta'):
        for a, b in zip(st[k], ref['st'][k]):
            assert torch.equal(a, b), k


@pytest.mark.parametrize('dtype', ['bf16', 'f32'])
def test_backward_finalize_folded_into_the_upsample_adjoint(dtype):
    gen = torch.Generator().manual_seed(14)
    keep = U.Keep()
    N, Cc, h, w, gstart = 3, 32, 14, 18, [0, 1, 3]
    G, ns = 2, L.STAT_SLOTS_FOLD
    t = U.rnd(torch.randn(N, Cc, h, w, generator=gen), dtype)
    g2 = U.rnd(torch.randn(N, Cc, 2 * h, 2 * w, generator=gen), dtype)
    st = _bn_state(keep, G, Cc, False, gen)
    bst = _random_slots(G, Cc, ns, gen, scale=10.0)
    mean = keep(U.fdev(0.2 * torch.randn(G, Cc, generator=gen)))
    invstd = keep(U.fdev(0.5 + torch.rand(G, Cc, generator=gen)))
    counts = [(gstart[k + 1] - gstart[k]) * 4 * h * w for k in range(G)]
    qd, PQR = _bwd_desc(keep, bst, st, mean, invstd, Cc, counts, ns)
    td, g2d = keep(U.nhwc(t, dtype)), keep(U.nhwc(g2, dtype))
    dt = torch.full((N, h, w, Cc), float('nan'), dtype=U.DT[dtype][1], device=U.dev())
    gs = L.gstart_array(gstart)
    lib = L.lib()
    snap = _snapshot(st)
    L.check(lib.rd_bn_finalize_bwd(C.byref(qd), None), 'finalize')
    L.check(lib.rd_up_bwd(L.ptr(g2d), L.ptr(td), L.ptr(dt), L.ptr(PQR[0]), L.ptr(PQR[1]), L.ptr(PQR[2]), N, h, w, Cc, G, gs, U.DT[dtype][0],
                          None, 0, None), 'up_bwd')
    torch.cuda.synchronize()
    ref = (dt.clone(), [x.clone() for x in PQR], _snapshot(st))
    _restore(st, snap)
    for x in PQR:
        x.fill_(float('nan'))
    dt.fill_(float('nan'))
    L.check(lib.rd_up_bwd(L.ptr(g2d), L.ptr(td), L.ptr(dt), L.ptr(PQR[0]), L.ptr(PQR[1]), L.ptr(PQR[2]), N, h, w, Cc, G, gs, U.DT[dtype][0],
                          _dev_copy(keep, qd), L.FIN_OWNER, None), 'up_bwd folded')
    torch.cuda.synchronize()
    assert torch.equal(dt, ref[0]) and torch.isfinite(dt.float()).all()
    for a, b in zip(PQR, ref[1]):
        assert torch.equal(a, b)
    for k in ('dgamma', 'dbeta'):
        for a, b in zip(st[k], ref[2][k]):
            assert torch.equal(a, b), k
