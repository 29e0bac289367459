"""Debug library only: shader-clock stamps of five workgroups (one per round of resident workgroups) of ONE conv_pf_kernel launch.
usage: RAMDSIR_DEBUG_LIB=1 python scripts/pf_trace.py [layer] [what]     (default: dec.convu2.conv3 dgrad, 64->64 at 200x200, 2560 tiles)
Events (thread 0 of the workgroup): start | chunk c: input tile in LDS, barrier, MFMAs done | last barrier | epilogue done."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import numpy as np, torch
from ramdsir import step as S, _lib
import bench as Bn
layer = sys.argv[1] if len(sys.argv) > 1 else 'dec.convu2.conv3'
what = sys.argv[2] if len(sys.argv) > 2 else 'dgrad'
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
sel = [op for op in OPS if len(op) > 2 and op[2].get('layer') == layer and op[2].get('what') == what]
assert len(sel) == 1, [(_op[2].get('layer'), _op[2].get('what')) for _op in OPS if len(_op) > 2][:400]
op = sel[0]
p = op[1][0]
p = p._obj if hasattr(p, '_obj') else p
raw = ctypes.CDLL(_lib.LIB_PATH)
tw, th = (25, 10) if (p.W % 25 == 0 or p.W <= 200) else (32, 8)
key = int(sys.argv[3]) if len(sys.argv) > 3 else ((p.W + tw - 1) // tw) * ((p.H + th - 1) // th)
assert raw.rd_debug_pf_trace(key, None) == 0
st = torch.cuda.current_stream()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
for _ in range(3):
    e0.record(st)
    assert op[0](*op[1], st.cuda_stream) == 0
    e1.record(st)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 16))()
assert raw.rd_debug_pf_trace(0, buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(8, 16).astype(np.int64)
print('%s %s: %dx%d, Cin %d, Cout %d, grid.x key %d, launch %.1f us' % (layer, what, p.H, p.W, p.Cin, p.Cout, key, e0.elapsed_time(e1) * 1e3))
w0 = t[0, 15]
print('wg   start(us, 100 MHz clock) | shader cycles since start: tile0 in LDS, barrier, MFMAs | tile1 in LDS, barrier, MFMAs | ... | last barrier, epilogue done')
for r in range(8):
    if t[r, 0] == 0:
        continue
    ev = [int(t[r, k] - t[r, 0]) if t[r, k] else -1 for k in range(15)]
    print('%4d  %8.1f | %s | end-barrier %6d  done %6d' % (7 + 600 * r, (t[r, 15] - w0) / 100.0, '  '.join('%6d %6d %6d' % tuple(ev[1 + 3 * c:4 + 3 * c]) for c in range(2) if ev[1 + 3 * c] >= 0), ev[13], ev[14]),
          '| epilogue: sums zeroed %6d  loads issued %6d  first vector stored %6d  all stored %6d  sums reduced %6d' % (ev[7], ev[8], ev[11], ev[9], ev[10]))
