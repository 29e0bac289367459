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
torch.manual_seed(0)
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, torch.bfloat16, bs, Sz, Sz, dataset='fundus', consistency='kd', lr=2e-3, total_iters=1000, ram='u8')
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(sum(bs), Sz, 0, 'cuda:0')
ts.load_raw(src, trg, lam); ts.load_target(mask)
for _ in range(3):
    ts.step()
torch.cuda.synchronize()
saved = ts._snapshot()


def run():
    ts._restore(saved)
    torch.cuda.synchronize()
    ts.step()
    torch.cuda.synchronize()
    return dict(params=bank.params.clone(), grads=bank.grads.clone(), losses=ts.losses.clone(), rec=ts.rec_mse.clone(),
                seg_stats=ts.seg.stat_arena.clone(), rec_stats=ts.rec.stat_arena.clone(),
                buffers=torch.cat([v.flatten().double() for v in bank.buffers.values()]))


ref = run()
bad = 0
for r in range(reps):
    cur = run()
    diff = [k for k in ref if not torch.equal(cur[k].view(torch.uint8), ref[k].view(torch.uint8))]
    if diff:
        bad += 1
        if bad <= 5:
            msg = []
            for k in diff:
                a, b = cur[k].flatten().double(), ref[k].flatten().double()
                idx = (a != b).nonzero().flatten()
                msg.append('%s: %d of %d values, first at %d (%.9g vs %.9g)' % (k, idx.numel(), a.numel(), int(idx[0]), float(a[idx[0]]), float(b[idx[0]])))
            print('repetition %d differs: %s' % (r, '; '.join(msg)), flush=True)
print('%d repetitions of one %dx%d step: %d differ from the first' % (reps, Sz, Sz, bad))
