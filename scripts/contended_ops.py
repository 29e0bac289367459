"""Every launch of one step timed twice: ALONE (one stream, scripts/layer_bench.py's method) and IN THE STEP (three streams, HIP events
recorded on the stream the launch goes to): which main-lane launches stretch most while the other lanes run beside them.
Events between all launches add marker packets, so the contended step is slower than the untimed one; the RATIOS are the result.
usage: contended_ops.py [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S
import bench as Bn

bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], 400, 400, ram='u8')
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(8, 400, 0, 'cuda:0')
ts.load_raw(src, trg, lam); ts.load_target(mask)
for _ in range(3):
    ts.step()
torch.cuda.synchronize()
main = torch.cuda.current_stream()


def timed(lanes, reps=3):
    acc = {}
    for _ in range(reps):
        ts.zero()
        evs = []

        def wrap(op, stream, launch):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            launch()
            e1.record(stream)
            evs.append((id(op), op, stream, e0, e1))
        ts.run_segment(ts.seg_a + ts.seg_b, main, lanes, wrap)
        ts.run_segment(ts.seg_c, main, lanes, wrap)
        torch.cuda.synchronize()
        for k, (oid, op, st, e0, e1) in enumerate(evs):
            acc.setdefault(k, [op, st, 0.0])[2] += e0.elapsed_time(e1) * 1e3 / reps
    return acc


alone = timed({})
cont = timed(ts.lanes())
lanes = ts.lanes()
names = {id(st): nm for nm, st in lanes.items()}
rows = []
for k in sorted(cont):
    op, st, us = cont[k]
    meta = op[2] if len(op) > 2 and op[2] else {}
    lane = names.get(id(st), 'main')
    rows.append((us - alone[k][2], us, alone[k][2], lane, op[0].__name__, meta.get('what', ''), meta.get('layer', '')))
tot = {}
for d, us, a, lane, fn, what, layer in rows:
    t = tot.setdefault(lane, [0.0, 0.0, 0])
    t[0] += us; t[1] += a; t[2] += 1
for lane, (us, a, n) in tot.items():
    print('lane %-6s %3d launches: %7.0f us in the step, %7.0f us alone (x%.2f)' % (lane, n, us, a, us / a))
byfn = {}
for d, us, a, lane, fn, what, layer in rows:
    if lane != 'main':
        continue
    t = byfn.setdefault(fn + ' ' + what, [0.0, 0.0, 0])
    t[0] += us; t[1] += a; t[2] += 1
print('main lane by entry point:')
for k, (us, a, n) in sorted(byfn.items(), key=lambda kv: -(kv[1][0] - kv[1][1])):
    print('  %-34s %3d launches: %7.0f us in the step, %7.0f alone (+%5.0f, x%.2f)' % (k, n, us, a, us - a, us / max(a, 1e-9)))
print('main-lane launches that stretch most:')
for d, us, a, lane, fn, what, layer in sorted([r for r in rows if r[3] == 'main'], reverse=True)[:int(sys.argv[1]) if len(sys.argv) > 1 else 25]:
    print('  +%6.1f us  %6.1f vs %6.1f alone  %-20s %-6s %s' % (d, us, a, fn, what, layer))
