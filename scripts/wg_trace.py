"""Debug library only: shader-clock stamps of workgroup 5 of ONE wgrad_ws_kernel launch, per tile: the loader wave's fill
(start, set consumed + LDS written, next set requested, barrier passed) and the MFMA wave's (arrival at the barrier, start, done).
usage: RAMDSIR_DEBUG_LIB=1 python scripts/wg_trace.py [layer]      (default dec.convu3.conv3: 128 -> 128 at 100x100)"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import numpy as np, torch
from ramdsir import step as S, _lib
import bench as Bn
layer = sys.argv[1] if len(sys.argv) > 1 else 'dec.convu3.conv3'
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], 400, 400, ram=True)
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(8, 400, 0, 'cuda:0')
ts.load_raw(src, trg, lam); ts.load_target(mask)
for _ in range(2):
    ts.zero(); ts.run_eager()
torch.cuda.synchronize()
OPS = [op for op in (ts.seg_a + ts.seg_b + ts.seg_c) if op[0] is not None]
sel = [op for op in OPS if len(op) > 2 and op[2].get('layer') == layer and op[2].get('kernel') == 'wgrad']
assert len(sel) == 1, len(sel)
op = sel[0]
p = op[1][0]._obj
tiles = p.N * ((p.H + 3) // 4) * ((p.W + 31) // 32)
raw = ctypes.CDLL(_lib.LIB_PATH)
assert raw.rd_debug_wg_trace(tiles, None) == 0
st = torch.cuda.current_stream()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for _ in range(3):
    e0.record(st)
    assert op[0](*op[1], st.cuda_stream) == 0
    e1.record(st)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2 * 64 * 4))()
assert raw.rd_debug_wg_trace(0, buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(2, 64, 4).astype(np.int64)
print('%s wgrad: %dx%d, Cin %d, Cout %d, N %d, %d tiles, cu_limit %d, rd_wgrad %.1f us (MFMA kernel + reduce)' % (layer, p.H, p.W, p.Cin, p.Cout, p.N, tiles, p.cu_limit, e0.elapsed_time(e1) * 1e3))
t0 = t[1, 0, 0]
print(' it | loader: start  +consume(wait, transform, LDS)  +request  +barrier | MFMA wave: at barrier  +wait  +MFMAs     (shader cycles)')
for i in range(24):
    l, m = t[1, i], t[0, i]
    if l[0] == 0:
        break
    print('%3d | %8d %7d %7d %7d | %8d %6d %6d' % (i, l[0] - t0, l[1] - l[0], l[2] - l[1], l[3] - l[2], m[0] - t0, m[1] - m[0], m[2] - m[1]))
print('MFMA waves: loop done at %d, partial block stored at %d' % (t[0, 63, 0] - t0, t[0, 63, 1] - t0))
