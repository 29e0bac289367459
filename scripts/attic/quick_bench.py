import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
Ssz = int(sys.argv[2]) if len(sys.argv) > 2 else 400
graph = (len(sys.argv) <= 3) or sys.argv[3] == 'graph'
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
torch.manual_seed(1337)
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
# kaiming-like init
for (m, k), (off, shape) in bank.index.items():
    v = bank.p(m, k)
    if len(shape) == 4:
        v.normal_(0, (2.0 / (shape[0] * shape[2] * shape[3])) ** 0.5)
    elif '.bn' in k and k.endswith('weight'):
        v.fill_(1.0)
    else:
        v.zero_()
ts = S.TrainStep(bank, mods, dtype, [2, 3, 3], Ssz, Ssz, dataset='fundus', consistency='kd',
                 options=dict(side_cus=0, rec_cus=0) if graph else None)
ts.wpack.refresh()
img = torch.rand(8, 3, Ssz, Ssz, device='cuda') * 2 - 1
imgf = (img + 0.2 * torch.randn_like(img)).clamp(-1, 1)
mask = (torch.rand(8, 2, Ssz, Ssz, device='cuda') > 0.5).float()
ts.load_images(img, imgf); ts.load_target(mask)
torch.cuda.synchronize()
if graph:
    ts.capture()
for _ in range(3):
    ts.step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    ts.step()
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print('dtype', dtype, 'S', Ssz, 'graph', graph, 'ms/step %.3f' % (dt * 1e3), 'img/s %.1f' % (8 / dt), 'losses', ts.loss_dict())
print('mem GB', torch.cuda.max_memory_allocated() / 1e9, 'nops', len(ts._ops))
