"""Debug library: entry / exit time of every workgroup of ONE conv_ws_kernel launch (100 MHz clock) -- the launch alone on the device and
the same launch inside the three-lane step: do workgroups start late when the lanes oversubscribe the CUs, or do they all run slower?
usage: RAMDSIR_DEBUG_LIB=1 RD_CONV_WS_TRACE_MIN=2560 RD_CONV_WS_TRACE_MODE=2 RD_CONV_WS_TRACE_CIN=64 python scripts/r6/wg_times.py"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import numpy as np, torch
from ramdsir import step as S, _lib
import bench as Bn
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], 400, 400, ram='u8')
ts.wpack.refresh()
src, trg, lam, mask, _ = Bn.synth_inputs(8, 400, 0, 'cuda:0')
ts.load_raw(src, trg, lam); ts.load_target(mask)
for dst, val in zip(ts.raw_slots[1], (src, trg, lam)):
    dst.copy_(val)
raw = ctypes.CDLL(_lib.LIB_PATH)


def dump(tag):
    buf = (ctypes.c_ulonglong * 512)()
    assert raw.rd_debug_ws_wg(buf) == 0
    t = np.array(buf, dtype=np.uint64).reshape(256, 2).astype(np.int64)
    t = t[t[:, 1] > 0]
    if not len(t):
        print(tag, 'no traced launch'); return
    t0 = t[:, 0].min()
    start, dur, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t[:, 0]) / 100.0, (t[:, 1] - t0) / 100.0
    q = lambda a, p: float(np.percentile(a, p))
    print('%-28s %3d workgroups: kernel %.1f us; start after the first: median %.1f  p90 %.1f  max %.1f us; own duration: median %.1f  p10 %.1f  p90 %.1f  max %.1f us; late starters (> 5 us): %d'
          % (tag, len(t), end.max(), q(start, 50), q(start, 90), start.max(), q(dur, 50), q(dur, 10), q(dur, 90), dur.max(), int((start > 5).sum())))


for _ in range(3):
    ts.reuse_next(); ts.step()
torch.cuda.synchronize()
for rep in range(4):
    ts.reuse_next(); ts.step()
    torch.cuda.synchronize()
    dump('in the step (rep %d)' % rep)
# the same launch alone
OPS = [op for op in (ts.seg_a + ts.seg_b + ts.seg_c) if op[0] is not None and len(op) > 2 and op[0].__name__ == 'rd_conv']
st = torch.cuda.current_stream()
mode = {'1': 'fwd', '2': 'dgrad'}.get(os.environ.get('RD_CONV_WS_TRACE_MODE', ''), None)
for op in OPS:
    if mode and op[2].get('what') != mode:
        continue
    before = (ctypes.c_ulonglong * 512)()
    raw.rd_debug_ws_wg(before)
    torch.cuda.synchronize()
    assert op[0](*op[1], st.cuda_stream) == 0
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 512)()
    raw.rd_debug_ws_wg(buf)
    if list(buf) != list(before):                            # this launch was a traced one
        dump('alone: %s %s' % (op[2].get('what'), op[2].get('layer')))
