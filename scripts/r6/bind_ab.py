"""Lane forks from bound dispatch-packet events (rd_run_list_bind_fork_events 1, the default) against recorded events (0): the pipelined
step timed alternately in ONE process, and the state after 6 steps compared bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S, _lib as L
import bench as Bn
bs, Sz = [2, 3, 3], int(sys.argv[1]) if len(sys.argv) > 1 else 400


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
    return bank, ts


def pipelined(ts):
    ts.reuse_next()
    ts.step()


state = {}
for mode in (0, 1):
    L.lib().rd_run_list_bind_fork_events(mode)
    bank, ts = make()
    for _ in range(6):
        pipelined(ts)
    torch.cuda.synchronize()
    state[mode] = (bank.params.clone(), bank.exp_avg_sq.clone(), ts.losses.clone(), ts.wpack.packed.clone())
same = all(torch.equal(a.reshape(-1).view(torch.uint8), b.reshape(-1).view(torch.uint8)) for a, b in zip(state[0], state[1]))
print('6 steps, recorded vs bound fork events: %s' % ('IDENTICAL' if same else 'DIFFER'))
assert same
for rnd in range(4):
    for mode in (0, 1):
        L.lib().rd_run_list_bind_fork_events(mode)
        for _ in range(10):
            pipelined(ts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            pipelined(ts)
        torch.cuda.synchronize()
        print('round %d bind_fork_events=%d %.3f ms/step' % (rnd, mode, (time.perf_counter() - t0) * 10))
L.lib().rd_run_list_bind_fork_events(1)
import ctypes
b, r = ctypes.c_longlong(0), ctypes.c_longlong(0)
L.lib().rd_run_list_fork_counts(ctypes.byref(b), ctypes.byref(r))
print('forks served by a bound event: %d, by a recorded event: %d' % (b.value, r.value))
