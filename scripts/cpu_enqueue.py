"""How long the CPU takes to ENQUEUE one eager step (3 streams, ctypes launches) next to how long the GPU takes to run it."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S
import bench as Bn
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], 400, 400, ram='u8')
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(8, 400, 0, 'cuda:0')
ts.load_raw(src, trg, lam); ts.load_target(mask)
for _ in range(5):
    ts.step()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    ts.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('eager: CPU enqueue %.2f ms/step, until GPU done %.2f ms/step (%d launches/step)' % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, len(ts._ops)))
# one step enqueued on an idle GPU: pure enqueue cost
torch.cuda.synchronize()
t0 = time.perf_counter(); ts.step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('single step on an idle GPU: enqueue %.2f ms, done after %.2f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
# the same step entry by entry from Python (Plan.run_lanes: one ctypes call + event record / stream wait per entry), for comparison
def python_step():
    ts.zero()
    ts.run_segment(ts.seg_a + ts.seg_b)
    ts.run_segment(ts.seg_c)
for _ in range(3):
    python_step()
torch.cuda.synchronize()
best = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); python_step(); t1 = time.perf_counter(); torch.cuda.synchronize()
    best.append((t1 - t0) * 1e3)
print('python launch loop, single step on an idle GPU: enqueue %.2f ms (min of 5: %.2f)' % (sum(best) / 5, min(best)))
best = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ts.step(); t1 = time.perf_counter(); torch.cuda.synchronize()
    best.append((t1 - t0) * 1e3)
print('native launch list (rd_run_list), single step on an idle GPU: enqueue %.2f ms (min of 5: %.2f)' % (sum(best) / 5, min(best)))
ts.launch_threads = True
for _ in range(5):
    ts.step()
torch.cuda.synchronize()
best = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ts.step(); t1 = time.perf_counter(); torch.cuda.synchronize()
    best.append((t1 - t0) * 1e3)
print('native launch list + lane worker threads (rd_run_list_threads), single step on an idle GPU: enqueue %.2f ms (min of 5: %.2f)' % (sum(best) / 5, min(best)))
t0 = time.perf_counter()
for _ in range(n):
    ts.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('... back to back: CPU enqueue %.2f ms/step, until GPU done %.2f ms/step' % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
ts.launch_threads = False
# one chain: rebuilt without the lane budgets (TrainStep.capture refuses them)
ts = S.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], 400, 400, ram='u8', options=dict(side_cus=0, rec_cus=0))
ts.wpack.refresh(); ts.load_raw(src, trg, lam); ts.load_target(mask)
ts.capture()
for _ in range(3):
    ts.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    ts.step()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('hipGraph (one chain): %.2f ms/step' % ((t2 - t0) / n * 1e3))
