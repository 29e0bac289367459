"""When does each lane of the eager 3-stream step finish?  (which lane the tail before Adam waits for)"""
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
acc = {}
n = 10
for it in range(n):
    ev = {k: torch.cuda.Event(enable_timing=True) for k in ['start', 'main', 'end'] + list(lanes)}
    ts.zero()
    ev['start'].record(main)
    used = E.Plan.run_lanes(ts.seg_a + ts.seg_b, main, lanes)
    ev['main'].record(main)
    for k in lanes:
        ev[k].record(lanes[k])
    for k in used:
        main.wait_stream(lanes[k])
    ts.run_segment(ts.seg_c, lanes=lanes)
    ev['end'].record(main)
    torch.cuda.synchronize()
    for k in ev:
        if k != 'start':
            acc[k] = acc.get(k, 0.0) + ev['start'].elapsed_time(ev[k]) / n
print('ms after step start (mean of %d steps): ' % n + '  '.join('%s %.2f' % (k, v) for k, v in sorted(acc.items(), key=lambda kv: kv[1])))
