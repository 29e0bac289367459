"""What a BatchNorm finalize costs between two dependent convs, alone on the GPU: a chain of N convs over two ping-pong tensors,
(A) no finalize at all (stale coefficients: timing only), (B) an explicit rd_bn_finalize_fwd launch after every conv, (C) the finalize
folded into the next conv's prologue (rd_src_t.fin), and -- debug library, RD_FIN_EXP bits in the high half of fin_flags -- the
folded form with phases switched off.  usage: fin_ubench.py [reps]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import torch
from ramdsir import _lib as L
import gpu_util as U
from test_gpu_ops import _conv_desc
from test_gpu_fold import _bn_state, _fwd_desc, _dev_copy

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
CHAIN = 24
CASES = [('small_fwd 16->16 @400 x16', 16, 16, 400), ('small_fwd 32->32 @200 x16', 32, 16, 200), ('ws 64->64 @100 x16', 64, 16, 100),
         ('ws 128->128 @50 x16', 128, 16, 50), ('pp 256->256 @25 x16', 256, 16, 25)]
lib = L.lib()
gen = torch.Generator().manual_seed(0)
for name, Cc, N, S in CASES:
    keep = U.Keep()
    gstart, G, ns = [0, N // 2, N], 2, L.STAT_SLOTS_FOLD
    bufs = [torch.randn(N, S, S, Cc, generator=gen).to(torch.bfloat16).to(U.dev()) for _ in range(2)]
    st = _bn_state(keep, G, Cc, True, gen)
    counts = [N // 2 * S * S] * 2
    w = torch.randn(Cc, Cc, 3, 3, generator=gen) / np.sqrt(9 * Cc)
    descs = []
    for k in range(2):                                      # conv k reads bufs[k] (BatchNorm of the previous conv pending), writes bufs[1 - k]
        stats = torch.zeros(G, L.STAT_SLOTS, Cc, 2, dtype=torch.float64, device=U.dev())
        stats[:, :ns, :, 0] = 1.0
        stats[:, :ns, :, 1] = 1e4
        keep(stats)
        fd, fb = _fwd_desc(keep, stats, st, Cc, counts, None, ns)
        for t in fb.values():
            t.fill_(0.5)
        descs.append((stats, fd, fb))
    convs = []
    for k in range(2):
        stats_in, fd_in, fb_in = descs[k]
        src = L.RdSrc()
        src.ptr, src.scale, src.shift = bufs[k].data_ptr(), fb_in['scale'].data_ptr(), fb_in['shift'].data_ptr()
        src.mode, src.C, src.slope, src.g_fixed = L.SRC_AFFACT, Cc, 0.0, -1
        p = _conv_desc(keep, [src], w, None, N, S, S, gstart, 'bf16', 9)
        p.emode, p.out, p.stats, p.stat_slots = 0, bufs[1 - k].data_ptr(), descs[1 - k][0].data_ptr(), ns
        convs.append((p, fd_in, _dev_copy(keep, fd_in)))

    def chain(mode, exp=0):
        for i in range(CHAIN):
            p, fd, fin = convs[i & 1]
            if mode == 'C':
                p.src[0].fin, p.src[0].fin_flags = fin, L.FIN_OWNER
            else:
                p.src[0].fin, p.src[0].fin_flags = None, 0
            if mode == 'B':
                L.check(lib.rd_bn_finalize_fwd(C.byref(fd), None), 'fin')
            L.check(lib.rd_conv(C.byref(p), L.RD_BF16, None), name)

    def timed(mode, exp=0):
        chain(mode, exp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            chain(mode, exp)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (reps * CHAIN) * 1e6

    variants = [('A none', 'A', 0), ('B explicit', 'B', 0), ('C folded', 'C', 0)]
    if os.environ.get('RAMDSIR_DEBUG_LIB') == '1':
        pass
    res = {}
    for r in range(3):
        for label, mode, exp in variants:
            res.setdefault(label, []).append(timed(mode, exp))
    base = sorted(res['A none'])[1]
    print('%-28s' % name + '  '.join('%s %.1f (%+.1f)' % (label, sorted(v)[1], sorted(v)[1] - base) for label, v in res.items()) + '  us per conv')
