"""Step time under TrainStep options, several configurations alive in ONE process and timed in alternating rounds (same box, same
clocks; the pipelined step as bench.py runs it).  usage: sweep_budgets.py "side_cus=96,rec_cus=128" "side_cus=160" ... [--rounds R] [--steps K]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S
import bench as Bn

args = [a for a in sys.argv[1:] if not a.startswith('--')]
rounds = int(sys.argv[sys.argv.index('--rounds') + 1]) if '--rounds' in sys.argv else 3
steps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 40
args = [a for a in args if not a.isdigit()]
specs = ['default'] + args
src, trg, lam, mask, _ = Bn.synth_inputs(8, 400, 0, 'cuda:0')
cfgs = []
for spec in specs:
    opt = {}
    if spec != 'default':
        for kv in spec.split(','):
            k, v = kv.split('=')
            opt[k] = int(v)
    bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
    Bn.init_weights(bank)
    ts = S.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], 400, 400, ram='u8', options=opt or None)
    ts.wpack.refresh()
    ts.load_raw(src, trg, lam); ts.load_target(mask)
    for dst, val in zip(ts.raw_slots[1], (src, trg, lam)):
        dst.copy_(val)
    cfgs.append((spec, ts))


def timed(ts):
    def fn():
        ts.reuse_next()
        ts.step()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


res = {s: [] for s, _ in cfgs}
for r in range(rounds):
    for spec, ts in cfgs:
        res[spec].append(timed(ts))
for spec, v in res.items():
    print('%-40s %s  median %.3f ms/step' % (spec, ' '.join('%.3f' % x for x in v), sorted(v)[len(v) // 2]), flush=True)
