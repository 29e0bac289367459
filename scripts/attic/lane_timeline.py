"""One three-stream step with a HIP event pair around every launch: start / end of each launch relative to the step's first launch,
per lane.  Shows what each lane is running at a given moment (e.g. what the weight-gradient lane still has to do when the main lane
reaches Adam).  The event packets slow the step by a few percent.  usage: lane_timeline.py [from_ms] [lane ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S
import bench as Bn

t_from = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
only = set(sys.argv[2:])
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
lanes = ts.lanes()
names = {id(st): nm for nm, st in lanes.items()}
best = None
for rep in range(3):
    ts.zero()
    evs = []
    t0 = torch.cuda.Event(enable_timing=True)
    t0.record(main)

    def wrap(op, stream, launch):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        launch()
        e1.record(stream)
        evs.append((op, stream, e0, e1))
    ts.run_segment(ts.seg_a + ts.seg_b, main, lanes, wrap)
    ts.run_segment(ts.seg_c, main, lanes, wrap)
    torch.cuda.synchronize()
    rows = [(t0.elapsed_time(e0), t0.elapsed_time(e1), names.get(id(st), 'main'), op) for op, st, e0, e1 in evs]
    end = max(r[1] for r in rows)
    if best is None or end < best[0]:
        best = (end, rows)
end, rows = best
print('step with per-launch events: %.3f ms' % end)
for lane in ['main'] + [n for n in lanes]:
    if only and lane not in only:
        continue
    print('lane %s' % lane)
    prev = None
    for s, e, ln, op in sorted(rows, key=lambda r: r[0]):
        if ln != lane or e < t_from:
            continue
        meta = op[2] if len(op) > 2 and op[2] else {}
        gap = '' if prev is None else ' (idle %5.1f us)' % ((s - prev) * 1e3)
        print('  %7.3f .. %7.3f  %6.1f us  %-22s %-12s %s%s' % (s, e, (e - s) * 1e3, op[0].__name__, meta.get('what', ''), meta.get('layer', ''), gap))
        prev = e
