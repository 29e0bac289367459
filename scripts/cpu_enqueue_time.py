import sys, os, time
ROOT='/root/repo'
sys.path[:0]=[ROOT, os.path.join(ROOT,'ram-dsir_amd')]
import torch
from ramdsir import step as S
import bench as Bn
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, torch.bfloat16, [2,3,3], 400, 400, ram=True)
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(8, 400, 0, 'cuda:0')
ts.load_raw(src, trg, lam); ts.load_target(mask)
for _ in range(3): ts.run_eager()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(10): ts.run_eager()
t1=time.perf_counter()
torch.cuda.synchronize()
t2=time.perf_counter()
print('cpu enqueue ms/step %.2f  total ms/step %.2f' % ((t1-t0)*100, (t2-t0)*100))
