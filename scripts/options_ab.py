"""A/B of TrainStep options in ONE process, alternating (same box, same clocks), on the bench workload (C2: 400 x 400, [2, 3, 3], bf16,
uint8 RAM inputs, pipelined stepping).  usage: options_ab.py [rounds] [steps] name=k:v,k:v ...   e.g.
    options_ab.py 5 40 explicit=fold_finalize:0 folded=fold_finalize:1"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S
import bench as Bn

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
variants = []
for spec in sys.argv[3:] or ['explicit=fold_finalize:0', 'folded=fold_finalize:1']:
    name, kv = spec.split('=', 1)
    variants.append((name, {k: int(v) for k, v in (item.split(':') for item in kv.split(',') if item)}))
size = int(os.environ.get('AB_SIZE', '400'))
steppers = []
for name, opts in variants:
    bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
    Bn.init_weights(bank)
    ts = S.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], size, size, ram='u8', options=opts)
    ts.wpack.refresh()
    src, trg, lam, mask, _ = Bn.synth_inputs(8, size, 0, 'cuda:0')
    ts.load_raw(src, trg, lam); ts.load_target(mask)
    for dst, val in zip(ts.raw_slots[1], (src, trg, lam)):
        dst.copy_(val)
    steppers.append((name, ts))


def timed(ts):
    for _ in range(5):
        ts.reuse_next(); ts.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ts.reuse_next(); ts.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


res = {}
for r in range(rounds):
    for name, ts in steppers:
        res.setdefault(name, []).append(timed(ts))
for name, ts in steppers:
    v = res[name]
    n_launch = sum(1 for op in ts._ops if op[0] is not None)
    print('%-20s %s  median %.3f ms/step  (%d launches in the list, loss %.4f)' % (name, ' '.join('%.3f' % x for x in v), sorted(v)[len(v) // 2], n_launch,
                                                                                  ts.loss_dict()['loss']))
