import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch, torch.distributed as dist
from ramdsir import streams as ST
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
if len(sys.argv) > 1 and sys.argv[1] == 'pg':
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29545')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
main = torch.cuda.current_stream(dev)
scratch = torch.zeros(256, device=dev)
cands = [torch.cuda.Stream(device=dev) for _ in range(8)]
def dt(busy, probe):
    torch.cuda.synchronize()
    with torch.cuda.stream(busy):
        torch.cuda._sleep(ST._SPIN)
    ev = torch.cuda.Event()
    with torch.cuda.stream(probe):
        scratch.add_(1.0); ev.record(probe)
    t0 = time.perf_counter(); ev.synchronize(); d = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    return d
for rnd in range(3):
    print('round', rnd, 'probe-vs-main dt ms:', ['%.2f' % dt(main, c) for c in cands])
print('main-vs-cand (cand busy):', ['%.2f' % dt(c, main) for c in cands])
picked = ST.pick_lanes(2, dev, [main])
print('picked vs main:', ['%.2f' % dt(main, p) for p in picked], 'picked pair:', '%.2f' % dt(picked[0], picked[1]))
for rnd in range(2):
    print('again picked vs main:', ['%.2f' % dt(main, p) for p in picked])
print('verified (GPU-timed):', [ST.verified.get(id(p)) for p in picked], 'beside main again:', [ST.runs_beside(main, p, scratch) for p in picked])
