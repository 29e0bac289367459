"""What a family of launches costs THE STEP (three streams, as bench.py runs it): the step is timed with that family's launches
left out of the launch list.  The skipped kernels' outputs stay at the values of the last complete step (coefficients,
gradients, partials), so every remaining kernel runs on realistic data -- the results of the ablated steps are wrong on
purpose, only their timing is used.  Answers "how much of the step would go away if family X were free", i.e. the upper bound
of any optimisation of X, before the kernel work is done.
usage: ablate_step.py [size] [steps]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S
import bench as Bn

Sz = int(sys.argv[1]) if len(sys.argv) > 1 else 400
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], Sz, Sz, ram='u8')
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(8, Sz, 0, 'cuda:0')
ts.load_raw(src, trg, lam); ts.load_target(mask)
full = (list(ts.seg_a), list(ts.seg_b), list(ts.seg_c))


def name(op):
    return op[0].__name__ if op[0] is not None else 'sync'


def kind(op):
    meta = op[2] if len(op) > 2 else {}
    return meta.get('kernel', ''), meta.get('what', ''), meta.get('layer', '')


FAMILIES = {
    'none': lambda op: False,
    'wgrad (all 40 rd_wgrad)': lambda op: name(op) == 'rd_wgrad',
    'wgrad >=64ch only': lambda op: name(op) == 'rd_wgrad' and op[2].get('flops', 0) / max(op[2].get('bytes', 1), 1) > 200,
    'bn_finalize fwd+bwd (76)': lambda op: name(op) in ('rd_bn_finalize_fwd', 'rd_bn_finalize_bwd'),
    'bn_finalize fwd (38)': lambda op: name(op) == 'rd_bn_finalize_fwd',
    'bn_finalize bwd (38)': lambda op: name(op) == 'rd_bn_finalize_bwd',
    'bn_apply': lambda op: name(op) == 'rd_bn_apply',
    'rec lane (whole restoration branch)': lambda op: len(op) > 2 and op[2].get('lane') == 'rec',
    'conv_small fwd+dgrad': lambda op: name(op) == 'rd_conv' and kind(op)[0].startswith('conv_small'),
    'conv64 fwd+dgrad': lambda op: name(op) == 'rd_conv' and kind(op)[0] == 'conv_kernel<bf16,9,2>',
    'up_stats+up_bwd': lambda op: name(op) in ('rd_up_stats', 'rd_up_bwd'),
    'pool fwd+bwd': lambda op: name(op) in ('rd_pool_fwd', 'rd_pool_bwd'),
    'ram': lambda op: name(op) == 'rd_ram_mix',
}


def timed(pred):
    ts.seg_a, ts.seg_b, ts.seg_c = ([op for op in seg if op[0] is None or not pred(op)] for seg in full)
    n_skipped = sum(1 for seg in full for op in seg if op[0] is not None and pred(op))
    for _ in range(3):
        ts.run_eager()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ts.run_eager()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, n_skipped


ts.seg_a, ts.seg_b, ts.seg_c = full
for _ in range(3):
    ts.run_eager()
torch.cuda.synchronize()
base = None
for fam, pred in FAMILIES.items():
    ms, n = timed(pred)
    if base is None:
        base = ms
    print('%-40s skipped %3d launches: %.3f ms/step  (%+.3f ms)' % (fam, n, ms, ms - base), flush=True)
ms, _ = timed(FAMILIES['none'])
print('%-40s %.3f ms/step (repeat of the complete step)' % ('none', ms))
