"""rd_ram_mix alone, repeated on the same inputs (8 pairs of 400 x 400 uint8 images): the kept row bins, the column results and the two outputs
are compared bit for bit with the first run.  usage: ram_stress.py reps [burst]   (run several at once, or beside STRESS_NORAM=1
step_repeat_stress.py processes, to put the GPU under load).  RAM_STRESS_SPIN=1: this process only generates load (no comparison)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import _lib as _L
if os.environ.get('RD_LIB_OVERRIDE'):                       # experiment: this process loads another build of the library
    _L.LIB_PATH = os.environ['RD_LIB_OVERRIDE']
from ramdsir import ram as R
import bench as Bn
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
burst = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B, S = 8, 400
src, trg, lam, mask, _ = Bn.synth_inputs(B, S, 0, 'cuda:0')
m = R.RamMixer(B, S, S, torch.bfloat16, 'cuda:0')
x = torch.zeros(2 * B, S, S, 8, dtype=torch.bfloat16, device='cuda:0')
m.bind(src, trg, lam, x[:B], x[B:])
KP = (m.b + 1 + 3) // 4 * 4
n1 = 2 * B * 3 * S * KP * 2


# RAM_STRESS_INPROC=<family>: the aggressor launches (scripts/r6/aggressor.py's selection) run in THIS process on a second stream beside RAM
AGG = None
if os.environ.get('RAM_STRESS_INPROC'):
    from ramdsir import step as S_
    pats = os.environ['RAM_STRESS_INPROC'].split(',')
    bank_, mods_ = S_.make_bank('cuda:0', 3, 16, 2, 3)
    Bn.init_weights(bank_)
    ts_ = S_.TrainStep(bank_, mods_, torch.bfloat16, [2, 3, 3], 400, 400, ram=None)
    ts_.wpack.refresh()
    g_ = torch.Generator().manual_seed(1)
    ts_.load_images((torch.rand(8, 3, 400, 400, generator=g_) * 2 - 1).cuda(), (torch.rand(8, 3, 400, 400, generator=g_) * 2 - 1).cuda())
    ts_.load_target(mask)
    for _ in range(2):
        ts_.run_eager()
    torch.cuda.synchronize()
    ops_ = [op for op in (ts_.seg_a + ts_.seg_b + ts_.seg_c) if op[0] is not None]
    nm_ = lambda op: '%s %s %s %s' % (op[0].__name__, (op[2] if len(op) > 2 else {}).get('kernel', ''), (op[2] if len(op) > 2 else {}).get('what', ''), (op[2] if len(op) > 2 else {}).get('layer', ''))
    AGG = [op for op in ops_ if any(p_ in nm_(op) for p_ in pats)]
    AGG_ST = torch.cuda.Stream()
    print('in-process aggressor: %d launches on a second stream' % len(AGG), flush=True)


def run():
    m.ws.zero_()
    x.zero_()
    torch.cuda.synchronize()
    if AGG is not None:
        for _ in range(int(os.environ.get('RAM_STRESS_AGG_REPEAT', '1'))):
            for op in AGG:
                assert op[0](*op[1], AGG_ST.cuda_stream) == 0
    for _ in range(burst):
        m.run()
    torch.cuda.synchronize()
    return dict(rowspec=m.ws[:n1].clone(), colout=m.ws[n1:n1 + n1 // 2].clone(), x=x.clone())


ref = run()
bad = {k: 0 for k in ref}
nbad = 0
for r in range(reps):
    cur = run()
    d = [k for k in ref if not torch.equal(cur[k].view(torch.uint8), ref[k].view(torch.uint8))]
    if d:
        nbad += 1
        for k in d:
            bad[k] += 1
        if nbad <= 4:
            info = []
            for k in d:
                idx = (cur[k].flatten().float() != ref[k].flatten().float()).nonzero().flatten()
                info.append('%s %d values first at %d' % (k, idx.numel(), int(idx[0]) if idx.numel() else -1))
            print('repetition %d: %s' % (r, '; '.join(info)), flush=True)
print('%d repetitions of rd_ram_mix (burst %d): %d differ from the first  %s' % (reps, burst, nbad, bad))
