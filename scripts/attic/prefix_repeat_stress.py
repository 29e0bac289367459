"""Under load, which launch of the forward pass produces different bits?  The first K launches of the step's forward list (direct entry-point
calls on the current stream, no launch list), preceded by the reset, repeated `reps` times; every tensor of the plan compared with the
first repetition.  usage: prefix_repeat_stress.py K [reps] [side]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S, _lib as L
import bench as Bn
K = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000; Sz = int(sys.argv[3]) if len(sys.argv) > 3 else 64
bs = [2, 3, 3]
torch.manual_seed(0)
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, torch.bfloat16, bs, Sz, Sz, dataset='fundus', consistency='kd', lr=2e-3, total_iters=1000, ram=None)
ts.wpack.refresh()
g_ = torch.Generator().manual_seed(1)
ts.load_images((torch.rand(8, 3, Sz, Sz, generator=g_) * 2 - 1).cuda(), (torch.rand(8, 3, Sz, Sz, generator=g_) * 2 - 1).cuda())
ops = list(ts.seg_a[:K]) if K > 0 else list(ts.seg_a)
print('launches:', len(ops), [(op[2].get('kernel'), op[2].get('layer')) if (len(op) > 2 and isinstance(op[2], dict)) else '?' for op in ops][-3:])
st = torch.cuda.current_stream().cuda_stream
NOZERO = os.environ.get('STRESS_NOZERO') == '1'


NATIVE = os.environ.get('STRESS_NATIVE', '1') == '1'         # through rd_run_list (the launches are enqueued back to back) or one ctypes call each
ts.seg_prefix = ops
watch = [t for t in ts.seg.keep if torch.is_tensor(t)][:12]


def run():
    if NATIVE:
        ts.launch(('seg_prefix',) if NOZERO else ('zero', 'seg_prefix'), lanes={}, join=True)
    else:
        if not NOZERO:
            ts.zero()
        for op in ops:
            L.check(op[0](*op[1], st), 'op')
    torch.cuda.synchronize()
    return [t.clone() for t in watch]


ref = run()
bad = 0
for r in range(reps):
    cur = run()
    d = [i for i, (a, b) in enumerate(zip(cur, ref)) if not torch.equal(a.view(torch.uint8), b.view(torch.uint8))]
    if d:
        bad += 1
        if bad <= 3:
            print('repetition %d: tensors %s differ' % (r, d[:10]), flush=True)
print('K=%d: %d of %d repetitions differ' % (K, bad, reps))
