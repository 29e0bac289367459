"""Which tensor moves first between two builds of the library?  The golden-test workload (128 x 128, [2, 3, 3], bf16) stepped `steps` times;
after every step every tensor of the two plans (activations, gradients, coefficient vectors, statistic arenas), the gradient arena and
the parameters are hashed.    usage: ab_bits.py dump out.json [steps] [f32|bf16]   |   ab_bits.py cmp a.json b.json"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]

if sys.argv[1] == 'cmp':
    a, b = json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))
    for st, (ra, rb) in enumerate(zip(a, b)):
        diff = [k for k in ra if ra[k] != rb.get(k)]
        print('step %d: %d of %d tensors differ%s' % (st, len(diff), len(ra), (': first ' + ', '.join(diff[:12])) if diff else ''))
    sys.exit(0)

import torch
from ramdsir import step as S
import bench as Bn
out, steps = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3
dtype = torch.float32 if (len(sys.argv) > 4 and sys.argv[4] == 'f32') else torch.bfloat16
BS, SIDE = [2, 3, 3], 128
torch.manual_seed(0)
bank, mods = S.make_bank('cuda:0', 3, 16, 2, len(BS))
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, dtype, BS, SIDE, SIDE, dataset='fundus', consistency='kd', lr=2e-3, total_iters=1000, ram='u8')
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(sum(BS), SIDE, 0, 'cuda:0')
ts.load_raw(src, trg, lam)
ts.load_target(mask)
h = lambda t: hashlib.sha1(t.contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:12]
rows = []
for st in range(steps):
    ts.step()
    torch.cuda.synchronize()
    r = {}
    for name, plan in (('seg', ts.seg), ('rec', ts.rec)):
        for i, t in enumerate(plan.keep):
            if torch.is_tensor(t):
                r['%s.keep[%03d] %s %s' % (name, i, tuple(t.shape), str(t.dtype).replace('torch.', ''))] = h(t)
        r['%s.stat_arena' % name] = h(plan.stat_arena)
    r['grads'] = h(bank.grads)
    r['params'] = h(bank.params)
    r['losses'] = h(ts.losses)
    rows.append(r)
json.dump(rows, open(out, 'w'))
print('wrote', out, len(rows), 'steps,', len(rows[0]), 'tensors')
