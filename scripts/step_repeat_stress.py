"""Bitwise repeatability of ONE training step: the step is run from the same saved state `reps` times (64 x 64 by default: 2 ms per step);
parameters after the step, gradients, loss terms and the BatchNorm statistic arenas are compared bit for bit with the first run, and the
first repetition that differs reports WHICH of them moved and, for the statistic arenas, where.  usage: step_repeat_stress.py [reps] [side]
(debug library: RD_SW_NWV=4 selects the two-workgroups-per-CU form of conv_small_fwd_kernel)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S
import bench as Bn
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
Sz = int(sys.argv[2]) if len(sys.argv) > 2 else 64
bs = [2, 3, 3]
DETAIL = os.environ.get('STRESS_DETAIL') == '1'
HEAD_ONLY = os.environ.get('STRESS_HEAD') == '1'
torch.manual_seed(0)
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
NORAM = os.environ.get('STRESS_NORAM') == '1'               # no RAM kernels in the step: x is loaded once
ts = S.TrainStep(bank, mods, torch.bfloat16, bs, Sz, Sz, dataset='fundus', consistency='kd', lr=2e-3, total_iters=1000, ram=None if NORAM else 'u8')
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(sum(bs), Sz, 0, 'cuda:0')
if NORAM:
    g_ = torch.Generator().manual_seed(1)
    ts.load_images((torch.rand(sum(bs), 3, Sz, Sz, generator=g_) * 2 - 1).cuda(), (torch.rand(sum(bs), 3, Sz, Sz, generator=g_) * 2 - 1).cuda())
else:
    ts.load_raw(src, trg, lam)
ts.load_target(mask)
for _ in range(3):
    ts.step()
torch.cuda.synchronize()
saved = ts._snapshot()


def run():
    ts._restore(saved)
    torch.cuda.synchronize()
    if HEAD_ONLY:                                            # the reset and the forward / decoder-backward segment only: no encoder backward, no Adam
        ts.launch(ts.head_names(), join=True)
    else:
        ts.step()
    torch.cuda.synchronize()
    out = dict(params=bank.params.clone(), grads=bank.grads.clone(), losses=ts.losses.clone(), rec=ts.rec_mse.clone(),
               seg_stats=ts.seg.stat_arena.clone(), rec_stats=ts.rec.stat_arena.clone(),
               buffers=torch.cat([v.flatten().double() for v in bank.buffers.values()]))
    if DETAIL and not NORAM:                                 # the RAM workspace: kept row bins [2B][3][H][KP] and column results [B][3][H][KP] (complex64)
        m = ts.rams[0]
        KP = (m.b + 1 + 3) // 4 * 4
        n1 = 2 * m.B * 3 * m.H * KP * 2
        out['ram.rowspec [2B][3][H][%d][2]' % KP] = m.ws[:n1].clone()
        out['ram.colout [B][3][H][%d][2]' % KP] = m.ws[n1:n1 + n1 // 2].clone()
    if DETAIL:                                               # every activation / gradient / coefficient tensor of the two plans, in allocation order
        for name, plan in (('seg', ts.seg), ('rec', ts.rec)):
            for i, t in enumerate(plan.keep):
                if not torch.is_tensor(t):
                    continue
                out['%s.keep[%d] %s %s' % (name, i, tuple(t.shape), str(t.dtype).replace('torch.', ''))] = t.clone()
    return out


# The fp64 statistic ARENAS are scratch accumulators: what the step consumes of them is the sum over the slot copies rounded to fp32.  A
# slot value itself is a long chain of fp64 atomic adds of fp32 terms, exact -- hence independent of arrival order -- only while the terms
# of one slot span < 2^29; at 400 x 400 with 8 slot copies a handful of the 1.2 M values (sums of squares ~1e5) differ in their
# LAST fp64 bit from run to run when other processes perturb the arrival order (round 6: 8-24 values per repetition, every derived tensor
# -- coefficients, running statistics, gradients, parameters -- bit-identical).  They are reported, not counted.
ARENAS = ('seg_stats', 'rec_stats')
ref = run()
bad = 0
arena_only = 0
for r in range(reps):
    cur = run()
    diff = [k for k in ref if not torch.equal(cur[k].view(torch.uint8), ref[k].view(torch.uint8))]
    if diff and all(k in ARENAS for k in diff):
        arena_only += 1
        worst = max(float(((cur[k] - ref[k]).abs() / ref[k].abs().clamp_min(1e-300)).max()) for k in diff)
        assert worst < 1e-14, 'statistic arenas differ by more than the last fp64 bits: %g' % worst
        diff = []
    if diff:
        bad += 1
        if bad <= 5:
            msg = []
            for k in diff[:40]:
                a, b = cur[k].flatten().double(), ref[k].flatten().double()
                a, b = torch.nan_to_num(a, nan=1e30), torch.nan_to_num(b, nan=1e30)
                idx = (a != b).nonzero().flatten()
                msg.append('%s: %d of %d values, first at %d (%.9g vs %.9g)' % (k, idx.numel(), a.numel(), int(idx[0]), float(a[idx[0]]), float(b[idx[0]])))
                if k.startswith('ram.'):                     # which (plane, row) pairs: index = ((plane * H + y) * KP + kx) * 2 + re/im
                    KP_ = int(k.split('[')[-2].rstrip(']'))
                    rows_ = sorted(set((int(i) // (2 * KP_ * Sz), (int(i) // (2 * KP_)) % Sz) for i in idx.tolist()[:200000]))
                    kxs_ = sorted(set((int(i) // 2) % KP_ for i in idx.tolist()[:200000]))
                    msg.append('    %s: %d (plane, y) pairs, first %s; bins kx %s' % (k.split()[0], len(rows_), rows_[:12], kxs_[:44]))
                if k.startswith('seg.keep[1]') and cur[k].dim() == 4:
                    N_, H_, W_, C_ = cur[k].shape
                    pix = sorted(set((int(i) // (H_ * W_ * C_), (int(i) // (W_ * C_)) % H_, (int(i) // C_) % W_) for i in idx.tolist()))
                    msg.append('    wrong pixels (n, y, x) of %s: %s' % (k.split()[0], pix[:60]))
            print('repetition %d differs (%d tensors):\n  %s' % (r, len(diff), '\n  '.join(msg)), flush=True)
if arena_only:
    print('%d repetitions differ ONLY in last fp64 bits of the statistic arenas (scratch; every derived tensor identical)' % arena_only)
print('%d repetitions of one %dx%d step: %d differ from the first' % (reps, Sz, Sz, bad))
