"""Is the eager forward chain launch-gap bound?  The seg plan's forward launches (encoder + seg decoder, one chain on the main stream)
timed three ways: the native launch list (rd_run_list) back to back, the same list captured into ONE hipGraph and replayed, and the sum
of the launches' own durations (HIP events around each launch).  usage: fwd_graph_probe.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S, engine as E
import bench as Bn

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], 400, 400, ram='u8')
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(8, 400, 0, 'cuda:0')
ts.load_raw(src, trg, lam); ts.load_target(mask)
for _ in range(3):
    ts.step()
torch.cuda.synchronize()
ts.seg_fwd_only = [ts.zero_op()] + list(ts.seg.fwd)
main = torch.cuda.current_stream()


def wall(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


eager = wall(lambda: ts.launch(('seg_fwd_only',), lanes={}))
st = torch.cuda.Stream()
st.wait_stream(main)
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(st):
    ts.launch(('seg_fwd_only',), lanes={})
    st.synchronize()
    with torch.cuda.graph(g, stream=st):
        ts.launch(('seg_fwd_only',), main=st, lanes={})
main.wait_stream(st)
graph = wall(g.replay)
acc = []


def wrap(op, stream, launch):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream); launch(); e1.record(stream)
    acc.append((e0, e1))
ts.run_segment(ts.seg_fwd_only, main, {}, wrap)
torch.cuda.synchronize()
acc.clear()
ts.run_segment(ts.seg_fwd_only, main, {}, wrap)
torch.cuda.synchronize()
ksum = sum(a.elapsed_time(b) for a, b in acc) * 1e3
print('%d forward launches: eager native list %.1f us, one hipGraph %.1f us, sum of per-launch event durations %.1f us' % (len(ts.seg_fwd_only), eager, graph, ksum))
