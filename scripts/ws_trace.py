"""Debug library only: per-step shader-clock stamps of one workgroup of conv_ws_kernel (the last launch that took it).
usage: RAMDSIR_DEBUG_LIB=1 RD_CONV_WS=1 RD_CONV_WS_TRACE_MIN=2000 python scripts/ws_trace.py   (2800 tiles = dec.convu2.conv3 forward, 64->64 at 200x200)"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import numpy as np, torch
from ramdsir import step as S, _lib
import bench as Bn
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], 400, 400, ram=True)
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(8, 400, 0, 'cuda:0')
ts.load_raw(src, trg, lam); ts.load_target(mask)
for _ in range(2):
    ts.zero(); ts.run_eager()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2 * 64 * 4))()
raw = ctypes.CDLL(_lib.LIB_PATH)
assert raw.rd_debug_ws_trace(buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(2, 64, 4).astype(np.int64)
t0 = t[0, 0, 0]
print('step | MFMA wave: start  +mfma_issued  +epilogue  +barrier | loader wave: start  +items  +weights  +barrier   (shader cycles, relative)')
for s in range(24):
    c, l = t[0, s], t[1, s]
    print('%3d  | %8d %6d %6d %6d | %8d %6d %6d %6d' % (s, c[0] - t0, c[1] - c[0], c[2] - c[1], c[3] - c[2], l[0] - t0, l[1] - l[0], l[2] - l[1], l[3] - l[2]))
k = t[0, 63]
print('kernel body: %d shader ticks, %d realtime ticks (100 MHz) -> %.1f us, shader clock %.2f GHz' % (k[2] - k[0], k[3] - k[1], (k[3] - k[1]) / 100.0, (k[2] - k[0]) / max(k[3] - k[1], 1) / 10.0))
