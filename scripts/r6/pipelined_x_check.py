"""Single process, the PIPELINED step as bench.py runs it: every step mixes the next batch (RAM) on the restoration lane beside its encoder
backward.  The synthetic batch is the same in both input slots, so every mixed x must equal, bit for bit, the x mixed alone on an idle GPU.
usage: pipelined_x_check.py [steps] [config C2|C3|C5|F256]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S
import bench as Bn
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
cfg = Bn.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else 'C2']
bs, Sz, ds = list(cfg['bs']), cfg['size'], cfg['dataset']
bank, mods = S.make_bank('cuda:0', 3, 16, cfg['num_classes'], len(bs))
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, torch.bfloat16, bs, Sz, Sz, dataset=ds, consistency=cfg['consistency'], lr=cfg['lr'], total_iters=21200, num_classes=cfg['num_classes'],
                 ram='u8' if ds == 'fundus' else True)
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(sum(bs), Sz, 0, 'cuda:0', ds)
ts.load_raw(src, trg, lam)
ts.load_target(mask)
for dst, val in zip(ts.raw_slots[1], (src, trg, lam)):
    dst.copy_(val)
torch.cuda.synchronize()
ts.rams[0].run()                                             # the reference: RAM alone on an idle GPU
torch.cuda.synchronize()
ref = ts.xbufs[0].clone()
ts.step()                                                    # classical first step (mixes slot 0 at its head)
torch.cuda.synchronize()
assert torch.equal(ts.xbufs[0].view(torch.int16), ref.view(torch.int16)), 'RAM at the head of a step differs from RAM alone'
bad = 0
for i in range(steps):
    ts.reuse_next()
    ts.step()                                                # mixes the other slot beside the encoder backward, then flips the slots
    torch.cuda.synchronize()
    cur = ts.x_current()
    if not torch.equal(cur.view(torch.int16), ref.view(torch.int16)):
        bad += 1
        if bad <= 5:
            d = (cur.view(torch.int16) != ref.view(torch.int16))
            idx = d.flatten().nonzero().flatten()
            print('step %d: x mixed beside the backward differs in %d values, first at %d' % (i, idx.numel(), int(idx[0])), flush=True)
print('%d pipelined steps: the mixed input differs from RAM alone in %d' % (steps, bad))
