"""A/B of the step's RAM placement in ONE process, alternating (same box, same clocks): classical (RAM at the head of every step),
pipelined (the next batch's RAM during the previous step, on the restoration lane beside the encoder backward: TrainStep.load_raw_next),
and both with the RAM launches removed (what RAM costs at all).  usage: pipeline_ab.py [rounds] [steps]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S
import bench as Bn

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], 400, 400, ram='u8')
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(8, 400, 0, 'cuda:0')
ts.load_raw(src, trg, lam); ts.load_target(mask)
for dst, val in zip(ts.raw_slots[1], (src, trg, lam)):
    dst.copy_(val)


def classical():
    ts.load_raw  # (inputs stay in slot 0)
    ts._x_ready = ts._next_loaded = False
    ts._slot = 0
    ts.step()


def pipelined():
    ts.reuse_next()
    ts.step()


saved = {k: list(getattr(ts, k)) for k in ('seg_ram_s0', 'seg_ram_s1', 'seg_b_pf_s0', 'seg_b_pf_s1')}


def no_ram(on):
    """remove / restore every RAM launch (stale network input: timing only)"""
    for k, v in saved.items():
        setattr(ts, k, [op for op in v if on or op[0] is None or op[0].__name__ != 'rd_ram_mix'] if not on else list(v))


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


res = {}
for r in range(rounds):
    for name, fn, ram_on in (('classical', classical, True), ('pipelined', pipelined, True), ('classical, no RAM', classical, False),
                             ('pipelined, no RAM', pipelined, False)):
        no_ram(ram_on)
        res.setdefault(name, []).append(timed(fn))
no_ram(True)
for name, v in res.items():
    print('%-20s %s  median %.3f ms/step' % (name, ' '.join('%.3f' % x for x in v), sorted(v)[len(v) // 2]))
