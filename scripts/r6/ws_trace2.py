"""Debug library: per-step shader-clock stamps of workgroup 5 of ONE conv_ws_kernel launch picked by (what, layer), launched alone.
usage: RAMDSIR_DEBUG_LIB=1 RD_CONV_WS_TRACE_MIN=1 python scripts/r6/ws_trace2.py fwd|dgrad <layer> [steps shown]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import numpy as np, torch
from ramdsir import step as S, _lib
import bench as Bn
what, layer = sys.argv[1], sys.argv[2]
show = int(sys.argv[3]) if len(sys.argv) > 3 else 20
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
sel = [op for op in OPS if len(op) > 2 and op[2].get('what') == what and op[2].get('layer') == layer and op[0].__name__ == 'rd_conv']
assert len(sel) == 1, [(op[2].get('what'), op[2].get('layer')) for op in OPS if len(op) > 2][:400]
st = torch.cuda.current_stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); assert sel[0][0](*sel[0][1], st.cuda_stream) == 0; e1.record()
torch.cuda.synchronize()
print('%s %s alone: %.1f us (%s)' % (what, layer, e0.elapsed_time(e1) * 1e3, sel[0][2].get('kernel')))
buf = (ctypes.c_ulonglong * (2 * 64 * 4))()
raw = ctypes.CDLL(_lib.LIB_PATH)
assert raw.rd_debug_ws_trace(buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(2, 64, 4).astype(np.int64)
t0 = t[0, 0, 0]
print('step | MFMA wave: start  +mfma_issued  +epilogue  +barrier | loader wave: start  +items  +weights  +barrier   (shader cycles, relative)')
for s in range(show):
    c, l = t[0, s], t[1, s]
    print('%3d  | %8d %6d %6d %6d | %8d %6d %6d %6d' % (s, c[0] - t0, c[1] - c[0], c[2] - c[1], c[3] - c[2], l[0] - t0, l[1] - l[0], l[2] - l[1], l[3] - l[2]))
k = t[0, 63]
print('kernel body: %d shader ticks, %d realtime ticks (100 MHz) -> %.1f us, shader clock %.2f GHz' % (k[2] - k[0], k[3] - k[1], (k[3] - k[1]) / 100.0, (k[2] - k[0]) / max(k[3] - k[1], 1) / 10.0))
if hasattr(raw, 'rd_debug_ws_fine'):
    fb = (ctypes.c_ulonglong * (2 * 16 * 16))()
    assert raw.rd_debug_ws_fine(fb) == 0
    f = np.array(fb, dtype=np.uint64).reshape(2, 16, 16).astype(np.int64)
    names = (['start', 'pref+g0', 'grp5', 'grp11', 'grp17', 'flush?', 'nb0v0', 'nb0v1', 'nb1v0', 'nb1v1', 'zero', 'advance', 'barrier'],
             ['start', 'begin', 'item0', 'item1', 'item2', 'item3', 'item4', 'item5', 'weights', 'shift', 'barrier'])
    for role, nm in ((0, 'MFMA wave'), (1, 'loader wave')):
        print('%s, stamps inside a step (cycles since the previous stamp; each stamp costs a scalar-memory wait)' % nm)
        print('step | ' + ' '.join('%7s' % n for n in names[role]))
        for s in range(min(show, 12)):
            row, prev = [], None
            for k in range(len(names[role])):
                v = f[role, s, k]
                if v == 0 or (prev is not None and v < prev):
                    row.append('      -')
                    continue
                row.append('%7d' % (v - (prev if prev is not None else t0)))
                prev = v
            print('%3d  | ' % s + ' '.join(row))
    e = f[0, 15]
    print('before the first step: folded finalize %d cycles, tables %d, first buffer (loader prologue) %d; first step starts %d cycles after kernel entry'
          % (e[1] - e[0], e[2] - e[1], e[3] - e[2], t0 - e[0]))
