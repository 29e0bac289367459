"""Per-launch timing of one eager fused step (HIP events on the launch stream), sorted by cost."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S
sys.path.insert(0, ROOT)
import bench as Bn
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
Sz = int(sys.argv[2]) if len(sys.argv) > 2 else 400
bs = [2, 3, 3]
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, dtype, bs, Sz, Sz, ram=True)
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(8, Sz, 0, 'cuda:0')
ts.load_raw(src, trg, lam); ts.load_target(mask)
OPS = [op for op in (ts.seg_a + ts.seg_b + ts.seg_c) if op[0] is not None]      # fork/join markers dropped: one stream here
for _ in range(2):
    ts.run_eager()
torch.cuda.synchronize()
st = torch.cuda.current_stream()
reps = 3
acc = {}
for rep in range(reps):
    ts.zero()
    evs = []
    for i, op in enumerate(OPS):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st)
        assert op[0](*op[1], st.cuda_stream) == 0
        e1.record(st)
        evs.append((i, e0, e1))
    torch.cuda.synchronize()
    for i, e0, e1 in evs:
        acc[i] = acc.get(i, 0.0) + e0.elapsed_time(e1) * 1e3 / reps
rows = []
for i, op in enumerate(OPS):
    meta = op[2] if len(op) > 2 else {}
    name = op[0].__name__
    rows.append((acc[i], i, name, meta.get('kernel', ''), meta.get('what', ''), meta.get('layer', ''), meta.get('bytes', 0), meta.get('flops', 0)))
tot = sum(r[0] for r in rows)
print('total us (sum of per-op event times, includes launch gaps): %.0f' % tot)
by = {}
for r in rows:
    by[r[2] + ' ' + r[3]] = by.get(r[2] + ' ' + r[3], 0) + r[0]
for k, v in sorted(by.items(), key=lambda kv: -kv[1]):
    print('  %-45s %8.0f us %5.1f%%' % (k, v, 100 * v / tot))
rows.sort(reverse=True)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    gbs = r[6] / (r[0] * 1e-6) / 1e9 if r[6] else 0
    tf = r[7] / (r[0] * 1e-6) / 1e12 if r[7] else 0
    print('%8.1f us  #%3d %-14s %-22s %-6s %-20s  alg %6.0f GB/s  %6.1f TF/s' % (r[0], r[1], r[2], r[3], r[4], r[5], gbs, tf))
