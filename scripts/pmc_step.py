"""One eager fused step (single stream) for rocprofv3 counter passes: every launch appears once per repetition, in plan
order, so per-kernel counters can be joined with scripts/layer_bench.py's timing rows.  usage: pmc_step.py [bf16|fp32] [size] [reps]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S
import bench as Bn
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
Sz = int(sys.argv[2]) if len(sys.argv) > 2 else 400
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, dtype, [2, 3, 3], Sz, Sz, ram='u8')           # uint8 pixels, as bench.py's Fundus workload
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(8, Sz, 0, 'cuda:0')
ts.load_raw(src, trg, lam); ts.load_target(mask)
for _ in range(reps):
    ts.zero()
    ts.run_eager()
torch.cuda.synchronize()
print('done')
