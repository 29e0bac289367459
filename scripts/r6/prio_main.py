"""The step with the MAIN lane on a high-priority stream (side / restoration lanes at normal priority) against the default stream: when
a CU frees up inside the backward pass, whose waiting workgroups get it?  (scripts/r6/wg_times.py: 71-128 of 160 workgroups of the
200 x 200 gradient launches start 5-26 us late.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S
import bench as Bn
bs, Sz = [2, 3, 3], 400
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else (0, -1)
print('stream priority range (least, greatest):', lo, hi)


def make():
    torch.manual_seed(0)
    bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
    Bn.init_weights(bank)
    ts = S.TrainStep(bank, mods, torch.bfloat16, bs, Sz, Sz, dataset='fundus', consistency='kd', lr=2e-3, total_iters=1000, ram='u8')
    ts.wpack.refresh()
    src, trg, lam, mask, _ = Bn.synth_inputs(sum(bs), Sz, 0, 'cuda:0')
    ts.load_raw(src, trg, lam); ts.load_target(mask)
    for dst, val in zip(ts.raw_slots[1], (src, trg, lam)):
        dst.copy_(val)
    return ts


def timed(ts, n=100):
    for _ in range(10):
        ts.reuse_next(); ts.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ts.reuse_next(); ts.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ts0 = make()
torch.cuda.synchronize()
hs = torch.cuda.Stream(priority=-1)
with torch.cuda.stream(hs):
    ts1 = make()
    torch.cuda.synchronize()
print('lanes default main:', ts0.lane_layout()); print('lanes high-priority main:', ts1.lane_layout())
for rnd in range(3):
    print('round %d default-stream main %.3f ms/step' % (rnd, timed(ts0)))
    with torch.cuda.stream(hs):
        print('round %d high-priority main  %.3f ms/step' % (rnd, timed(ts1)))
