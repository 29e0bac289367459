"""Does the main lane WAIT at the restoration lane's join?  Events on the main stream right before the join (after the seg decoder's
backward) and on the restoration stream behind its last launch, both relative to the step's start; likewise for the final join of the
weight-gradient lane in front of Adam.  The lane that arrives second is the one the step waits for."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S, engine as E
import bench as Bn
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], 400, 400, ram='u8')
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(8, 400, 0, 'cuda:0')
ts.load_raw(src, trg, lam); ts.load_target(mask)
for _ in range(5):
    ts.step()
torch.cuda.synchronize()
main = torch.cuda.current_stream()
lanes = ts.lanes()
ops = ts.seg_a
j = max(i for i, op in enumerate(ops) if op[0] is None and op[1] == ('join', 'rec'))
f = min(i for i, op in enumerate(ops) if op[0] is None and op[1] == ('fork', 'rec'))
acc = {}
n = 10
for it in range(n):
    ev = {k: torch.cuda.Event(enable_timing=True) for k in ('start', 'fork', 'main_at_join', 'rec_done', 'after_join', 'main_end', 'side_end', 'end')}
    ts.zero()
    ev['start'].record(main)
    used = E.Plan.run_lanes(ops[:f], main, lanes)
    ev['fork'].record(main)
    used |= E.Plan.run_lanes(ops[f:j], main, lanes)
    ev['main_at_join'].record(main)
    ev['rec_done'].record(lanes['rec'])
    used |= E.Plan.run_lanes(ops[j:], main, lanes)
    ev['after_join'].record(main)
    used |= E.Plan.run_lanes(ts.seg_b, main, lanes)
    ev['main_end'].record(main)
    ev['side_end'].record(lanes['side0'])
    for k in used:
        main.wait_stream(lanes[k])
    ts.run_segment(ts.seg_c, lanes=lanes)
    ev['end'].record(main)
    torch.cuda.synchronize()
    for k in ev:
        if k != 'start':
            acc[k] = acc.get(k, 0.0) + ev['start'].elapsed_time(ev[k]) / n
print('ms after step start (mean of %d steps):' % n)
for k, v in sorted(acc.items(), key=lambda kv: kv[1]):
    print('  %-14s %.3f' % (k, v))
print('main waits for the restoration lane at the join: %.3f ms' % max(0.0, acc['rec_done'] - acc['main_at_join']))
print('main waits for the weight-gradient lane in front of Adam: %.3f ms' % max(0.0, acc['side_end'] - acc['main_end']))
