"""Load generator for the foreign-process corruption hunt: the launches of ONE family of the step (substring of `entry point + kernel + what`
of engine.py's meta; 'all' = every launch), repeated for `seconds` on one stream.  usage: aggressor.py <substring[,substring..]> [seconds]
A victim (scripts/r6/ram_stress.py) runs beside it in another process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S
import bench as Bn
pats = sys.argv[1].split(',')
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], 400, 400, ram=None)
ts.wpack.refresh()
g_ = torch.Generator().manual_seed(1)
ts.load_images((torch.rand(8, 3, 400, 400, generator=g_) * 2 - 1).cuda(), (torch.rand(8, 3, 400, 400, generator=g_) * 2 - 1).cuda())
_, _, _, mask, _ = Bn.synth_inputs(8, 400, 0, 'cuda:0')
ts.load_target(mask)
for _ in range(2):
    ts.run_eager()
torch.cuda.synchronize()
OPS = [op for op in (ts.seg_a + ts.seg_b + ts.seg_c) if op[0] is not None]
def name(op):
    meta = op[2] if len(op) > 2 else {}
    return '%s %s %s %s' % (op[0].__name__, meta.get('kernel', ''), meta.get('what', ''), meta.get('layer', ''))
sel = [op for op in OPS if pats == ['all'] or any(p in name(op) for p in pats)]
print('aggressor: %d of %d launches match %s' % (len(sel), len(OPS), pats), flush=True)
st = torch.cuda.current_stream()
t0, n = time.time(), 0
while time.time() - t0 < secs:
    for op in sel:
        assert op[0](*op[1], st.cuda_stream) == 0
    n += 1
    if n % 20 == 0:
        torch.cuda.synchronize()
torch.cuda.synchronize()
print('aggressor: %d rounds' % n)
