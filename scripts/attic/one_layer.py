"""Run ONE launch of the fused step (selected by layer name + fwd/dgrad/wgrad) many times, for rocprofv3 --pmc."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S
import bench as Bn
layer, what = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], 400, 400, ram=True)
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(8, 400, 0, 'cuda:0')
ts.load_raw(src, trg, lam); ts.load_target(mask)
ts.run_eager()
torch.cuda.synchronize()
st = torch.cuda.current_stream()
sel = [op for op in ts._ops if op[0] is not None and len(op) > 2 and op[2].get('layer') == layer and (op[2].get('what', 'wgrad') == what or (what == 'wgrad' and op[2].get('kernel') == 'wgrad'))]
assert len(sel) == 1, [o[2] for o in ts._ops if o[0] is not None and len(o) > 2][:5]
op = sel[0]
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(reps):
    assert op[0](*op[1], st.cuda_stream) == 0
e1.record(st)
torch.cuda.synchronize()
print(layer, what, 'avg us', e0.elapsed_time(e1) * 1e3 / reps, op[2])
