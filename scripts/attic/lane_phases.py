"""Main-lane milestones of the eager 3-stream step (ms after step start) next to the isolated sums of the same launches."""
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
ops = ts.seg_a + ts.seg_b
def meta(op):
    return op[2] if len(op) > 2 and isinstance(op[2], dict) else {}
names = [(getattr(op[0], '__name__', str(op[0])) if op[0] is not None else 'sync', meta(op)) for op in ops]
def first(pred, start=0):
    for i in range(start, len(ops)):
        if pred(*names[i]):
            return i
    return len(ops)
i_dec = first(lambda n, m: m.get('layer', '').startswith('dec.'))
i_loss = first(lambda n, m: n == 'rd_seg_loss')
i_encb = len(ts.seg_a)
cuts = [('encoder fwd (+RAM)', 0, i_dec), ('seg decoder fwd [rec branch beside]', i_dec, i_loss), ('seg loss', i_loss, i_loss + 1),
        ('seg decoder bwd [rec branch + wgrads beside]', i_loss + 1, i_encb), ('encoder bwd [wgrads beside]', i_encb, len(ops))]
main = torch.cuda.current_stream()
lanes = ts.lanes()
n = 10
acc = [0.0] * len(cuts)
for it in range(n):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(cuts) + 1)]
    ts.zero()
    evs[0].record(main)
    used = set()
    for k, (_, a, b) in enumerate(cuts):
        used |= E.Plan.run_lanes(ops[a:b], main, lanes)
        evs[k + 1].record(main)
    for k in used:
        main.wait_stream(lanes[k])
    ts.run_segment(ts.seg_c, lanes=lanes)
    torch.cuda.synchronize()
    for k in range(len(cuts)):
        acc[k] += evs[k].elapsed_time(evs[k + 1]) / n
for (label, a, b), t in zip(cuts, acc):
    nmain = sum(1 for op in ops[a:b] if op[0] is not None and meta(op).get('lane') is None and meta(op).get('kernel') != 'wgrad')
    print('%-50s %.2f ms on the main stream (%d launches in the segment, %d of them main-lane)' % (label, t, b - a, nmain))
