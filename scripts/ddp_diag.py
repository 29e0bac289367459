"""Why is the data-parallel eager step slow with some HW-queue layouts?  Forced DDP on one GPU (world 1)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch, torch.distributed as dist
from ramdsir import step as S, ddp as D
import bench as Bn
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
bank, mods = S.make_bank(dev, 3, 16, 2, 3)
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], 400, 400, ram='u8')
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(8, 400, 0, dev)
ts.load_raw(src, trg, lam); ts.load_target(mask)
runner = D.DataParallelStep(ts)
x = torch.zeros(1024, device=dev)
def independent(a, b):
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        torch.cuda._sleep(8_000_000)
    e = torch.cuda.Event()
    with torch.cuda.stream(b):
        x.add_(1.0); e.record(b)
    t0 = time.perf_counter(); e.synchronize(); dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt < 1e-3
def timeit(fn, label, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    print('%-50s %.3f ms/step' % (label, (time.perf_counter() - t0) / n * 1e3))
def conc(tag):
    S_ = {'main': torch.cuda.current_stream(), 'side': ts.side[0], 'rec': ts.rec_stream}
    print(tag, {a + '/' + b: int(independent(S_[a], S_[b])) for a in S_ for b in S_ if a < b}, {b + '/' + a: int(independent(S_[b], S_[a])) for a in S_ for b in S_ if a < b})
conc('after TrainStep creation:')
timeit(ts.step, 'plain eager step')
conc('after plain steps:')
timeit(runner.step, 'DataParallelStep.step (first, creates NCCL stream)')
S_ = {'main': torch.cuda.current_stream(), 'side': ts.side[0], 'rec': ts.rec_stream, 'comm': runner.comm}
print('pairwise concurrency:', {a + '/' + b: int(independent(S_[a], S_[b])) for a in S_ for b in S_ if a < b})
orig = runner.buckets.reduce
runner.buckets.reduce = lambda i, async_op=True: None
timeit(runner.step, 'DDP step without the all_reduce calls')
runner.buckets.reduce = orig
timeit(runner.step, 'DDP step')
timeit(ts.step, 'plain eager step again')
dist.destroy_process_group()
