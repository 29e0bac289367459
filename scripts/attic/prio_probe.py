"""Does a HIP stream priority for the critical-path lane change the eager 3-stream step time?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S
import bench as Bn
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else None)
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], 400, 400, ram='u8')
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(8, 400, 0, 'cuda:0')
ts.load_raw(src, trg, lam); ts.load_target(mask)
def run(label, main=None):
    ctx = torch.cuda.stream(main) if main is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(5):
            ts.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            ts.step()
        torch.cuda.synchronize()
        print('%-40s %.3f ms/step' % (label, (time.perf_counter() - t0) / 30 * 1e3))
run('default streams')
run('main lane on a priority -1 stream', torch.cuda.Stream(priority=-1))
old = (ts.side, ts.rec_stream)
ts.side = [torch.cuda.Stream(priority=-1) for _ in ts.side]
run('side (wgrad) lane priority -1')
ts.side = old[0]
ts.rec_stream = torch.cuda.Stream(priority=-1)
run('rec lane priority -1')
ts.rec_stream = old[1]
run('default again')
run('main on a fresh priority 0 stream', torch.cuda.Stream(priority=0))
