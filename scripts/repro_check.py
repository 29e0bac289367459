"""Is the step run-to-run reproducible?  One TrainStep, the same parameters / optimizer state / inputs restored before every run: the
parameter gradients, the updated parameters and the losses of R runs are compared BITWISE with those of the first (and by relative L2
where they differ).  usage: repro_check.py [f32|bf16] [size] [runs] [lanes: 3|1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import numpy as np
import torch
from ramdsir import step as S
import bench as Bn

dtype = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == 'f32') else torch.bfloat16
Sz = int(sys.argv[2]) if len(sys.argv) > 2 else 400
R = int(sys.argv[3]) if len(sys.argv) > 3 else 6
one_lane = len(sys.argv) > 4 and sys.argv[4] == '1'
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
opts = dict(fork=False, side_cus=0, rec_cus=0) if one_lane else None
ts = S.TrainStep(bank, mods, dtype, [2, 3, 3], Sz, Sz, ram=True, options=opts)
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(8, Sz, 0, 'cuda:0')
ts.load_raw(src, trg, lam); ts.load_target(mask)
for _ in range(2):
    ts.step()
torch.cuda.synchronize()
saved = ts._snapshot()
runs = []
for r in range(R):
    ts._restore(saved)
    ts.load_raw(src, trg, lam); ts.load_target(mask)
    torch.cuda.synchronize()
    ts.step()
    torch.cuda.synchronize()
    runs.append(dict(grads=bank.grads.clone(), params=bank.params.clone(), losses=ts.losses.clone(), rec=ts.rec_mse.clone()))
print('%s %dx%d, %d runs, %s' % ('f32' if dtype == torch.float32 else 'bf16', Sz, Sz, R, 'one stream' if one_lane else 'three streams'))
ref = runs[0]
for r in range(1, R):
    out = []
    for k in ('losses', 'rec', 'grads', 'params'):
        a, b = ref[k], runs[r][k]
        same = torch.equal(a, b)
        nd = int((a != b).sum())
        rel = float((a.double() - b.double()).norm() / (a.double().norm() + 1e-30))
        out.append('%s %s (%d of %d differ, rel L2 %.2e)' % (k, 'BITWISE' if same else 'differs', nd, a.numel(), rel))
    print('run %d vs run 0: ' % r + ' | '.join(out))
# per-tensor: which parameter gradients differ first (in backward order the decoders' last layers come first)
g0, g1 = ref['grads'], runs[1]['grads']
if not torch.equal(g0, g1):
    worst = []
    for (m, key), (off, shape) in bank.index.items():
        n = int(np.prod(shape)) if len(shape) else 1
        a, b = g0[off:off + n], g1[off:off + n]
        if not torch.equal(a, b):
            worst.append((float((a.double() - b.double()).norm() / (a.double().norm() + 1e-30)), m, key))
    worst.sort(reverse=True)
    print('%d of %d gradient tensors differ; largest relative L2:' % (len(worst), len(bank.index)))
    for w in worst[:8]:
        print('   %.2e  %s.%s' % w)
