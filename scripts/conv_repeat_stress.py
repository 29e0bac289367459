"""Bitwise repeatability of ONE forward launch of the <=16-channel class (conv_small_fwd_kernel) with a folded BatchNorm finalize in its
prologue: the same launch `reps` times on the same inputs, output tensor and output statistic slots compared bit for bit with the first
run.  usage: conv_repeat_stress.py [reps] [side] [slots]        (debug library: RD_SW_NWV = 4 / 8 selects the workgroup shape)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import torch
from ramdsir import _lib as L
import gpu_util as U
from test_gpu_ops import _conv_desc
from test_gpu_fold import _bn_state, _fwd_desc, _dev_copy

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ns = int(sys.argv[3]) if len(sys.argv) > 3 else L.STAT_SLOTS_FOLD
lib = L.lib()
gen = torch.Generator().manual_seed(0)
keep = U.Keep()
Cc, N = 16, 16
RAW = os.environ.get('STRESS_RAW') == '1'                   # the first conv of the network: a raw 8-channel-slot input, no BatchNorm in front
BURST = int(os.environ.get('STRESS_BURST', '1'))            # launches enqueued back to back (each into its own output tensor) per synchronisation
gstart, G = [0, 4, 10, 16], 3
x = torch.randn(N, S, S, 8 if RAW else Cc, generator=gen)
if os.environ.get('STRESS_SPAN') == '1':                    # rows of wildly different magnitude: partial sums of one channel span > 2^29
    x *= (10.0 ** (torch.arange(S).float() % 13 - 6)).view(1, S, 1, 1)
x = x.to(torch.bfloat16).to(U.dev())
st = _bn_state(keep, G, Cc, True, gen)
counts = [(gstart[i + 1] - gstart[i]) * S * S for i in range(G)]
stats_in = torch.zeros(G, L.STAT_SLOTS, Cc, 2, dtype=torch.float64, device=U.dev())
stats_in[:, :ns, :, 0] = torch.randn(G, ns, Cc, generator=gen, dtype=torch.float64).to(U.dev()) * 10
stats_in[:, :ns, :, 1] = 1e3 + torch.rand(G, ns, Cc, generator=gen, dtype=torch.float64).to(U.dev()) * 1e3
keep(stats_in)
fd, fb = _fwd_desc(keep, stats_in, st, Cc, counts, None, ns)
w = torch.randn(Cc, 3 if RAW else Cc, 3, 3, generator=gen) / np.sqrt(9 * Cc)
src = L.RdSrc()
src.ptr, src.scale, src.shift = x.data_ptr(), fb['scale'].data_ptr(), fb['shift'].data_ptr()
src.mode, src.C, src.slope, src.g_fixed = (L.SRC_RAW, 8, 1.0, -1) if RAW else (L.SRC_AFFACT, Cc, 0.0, -1)
p = _conv_desc(keep, [src], w, None, N, S, S, gstart, 'bf16', 9)
out = torch.empty(N, S, S, Cc, dtype=torch.bfloat16, device=U.dev())
stats_out = torch.zeros(G, L.STAT_SLOTS, Cc, 2, dtype=torch.float64, device=U.dev())
p.emode, p.out, p.stats, p.stat_slots = 0, out.data_ptr(), stats_out.data_ptr(), ns
fin = _dev_copy(keep, fd)
if not RAW:
    p.src[0].fin, p.src[0].fin_flags = fin, 0               # not the owner: no running-statistics update, the same inputs every time
ref = None
hashes = {}
bad_out = bad_stats = bad_slots = 0
outs = [torch.empty(N, S, S, Cc, dtype=torch.bfloat16, device=U.dev()) for _ in range(BURST)]
for r in range(0, reps, BURST):
    stats_out.zero_()
    for o in outs:
        o.zero_()
    for o in outs:
        p.out = o.data_ptr()
        L.check(lib.rd_conv(C.byref(p), L.RD_BF16, None), 'conv')
    torch.cuda.synchronize()
    for bi, o in enumerate(outs):
        cur_out = o.view(torch.int16)
        if ref is None:
            ref = (cur_out.clone(), None, None)
            continue
        if not torch.equal(cur_out, ref[0]):
            bad_out += 1
            if bad_out <= 3:
                idx = (cur_out != ref[0]).flatten().nonzero().flatten()
                pix = sorted(set((int(i) // (S * S * Cc), (int(i) // (S * Cc)) % S, (int(i) // Cc) % S) for i in idx.tolist()))
                print('launch %d (burst position %d): %d values differ; pixels (n, y, x): %s' % (r + bi, bi, idx.numel(), pix[:40]), flush=True)
print('%d launches at %dx%d, %d slots: output differs %d times, slot sums (fixed-order sum over the slots) %d times, individual slots %d times'
      % (reps, S, S, ns, bad_out, bad_stats, bad_slots))

