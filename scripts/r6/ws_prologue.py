"""Debug library: what every conv_ws_kernel launch of the step spends BEFORE its first step (workgroup 5, shader cycles): folded BatchNorm
finalize, descriptor / bias tables, the loader waves' prologue (first buffer), and the kernel body behind it.
usage: RAMDSIR_DEBUG_LIB=1 RD_CONV_WS_TRACE_MIN=1 python scripts/r6/ws_prologue.py"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
raw = ctypes.CDLL(_lib.LIB_PATH)
st = torch.cuda.current_stream()
OPS = [op for op in (ts.seg_a + ts.seg_b + ts.seg_c) if op[0] is not None and len(op) > 2 and op[0].__name__ == 'rd_conv']
print('%-6s %-20s %9s | %8s %8s %8s | %9s %9s  (us at the measured clock; alone = events around the launch)' % ('what', 'layer', 'alone us', 'finalize', 'tables', 'buffer0', 'body us', 'pre us'))
tot = dict(fwd=[0.0, 0.0, 0.0], dgrad=[0.0, 0.0, 0.0])
for op in OPS:
    what, layer = op[2].get('what'), op[2].get('layer')
    fb = (ctypes.c_ulonglong * (2 * 16 * 16))()
    z = (ctypes.c_ulonglong * (2 * 64 * 4))()
    # clear the stamps of the previous launch: a launch that is not on conv_ws_kernel leaves zeros
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); assert op[0](*op[1], st.cuda_stream) == 0; e1.record()
    torch.cuda.synchronize()
    assert raw.rd_debug_ws_fine(fb) == 0 and raw.rd_debug_ws_trace(z) == 0
    f = np.array(fb, dtype=np.uint64).reshape(2, 16, 16).astype(np.int64)
    t = np.array(z, dtype=np.uint64).reshape(2, 64, 4).astype(np.int64)
    e, k = f[0, 15], t[0, 63]
    if k[3] <= k[1] or e[3] <= e[0] or k[0] != e[2]:
        continue                                              # not a conv_ws_kernel launch (stamps of an earlier one)
    ghz = (k[2] - k[0]) / max(k[3] - k[1], 1) / 10.0
    us = lambda c: c / ghz / 1e3
    alone = e0.elapsed_time(e1) * 1e3
    print('%-6s %-20s %9.1f | %8.2f %8.2f %8.2f | %9.1f %9.2f' % (what, layer, alone, us(e[1] - e[0]), us(e[2] - e[1]), us(e[3] - e[2]), (k[3] - k[1]) / 100.0, us(e[3] - e[0])))
    if what in tot:
        tot[what][0] += alone; tot[what][1] += us(e[3] - e[0]); tot[what][2] += (k[3] - k[1]) / 100.0
for w, v in tot.items():
    print('%s launches: %.0f us alone, %.0f us before the first step, %.0f us of kernel body' % (w, v[0], v[1], v[2]))
