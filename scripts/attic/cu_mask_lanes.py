"""Experiment: the side lanes on streams with a COMPUTE-UNIT MASK (hipExtStreamCreateWithCUMask) instead of plain streams whose
persistent kernels are merely limited in workgroup count (tuning.py side_cus / rec_cus): does confining the weight-gradient lane and the
restoration lane to disjoint CU sets speed up the main lane?  usage: cu_mask_lanes.py  (prints ms/step per variant)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S, streams as ST
import bench as Bn

hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(bits):
    """bits: iterable of CU indices (0..255) the stream may use."""
    words = [0] * 8
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (ctypes.c_uint32 * 8)(*words)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def build(side_cus, rec_cus):
    bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
    Bn.init_weights(bank)
    ts = S.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], 400, 400, ram='u8', options=dict(side_cus=side_cus, rec_cus=rec_cus))
    ts.wpack.refresh()
    src, trg, lam, mask, _ = Bn.synth_inputs(8, 400, 0, 'cuda:0')
    ts.load_raw(src, trg, lam); ts.load_target(mask)
    return ts


def time_steps(ts, main=None, n=30):
    ctx = torch.cuda.stream(main) if main is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(5):
            ts.step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ts.step()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


variants = [
    ('baseline: plain streams, budgets 128 / 128', 128, 128, None, None, None),
    ('side mask [0,128)  rec mask [128,256)  main all', 128, 128, range(0, 128), range(128, 256), None),
    ('side mask [0,96)   rec mask [96,192)   main all', 96, 96, range(0, 96), range(96, 192), None),
    ('side mask [0,64)   rec mask [64,192)   main all', 64, 128, range(0, 64), range(64, 192), None),
    ('side mask [0,128)  rec mask [0,128)    main all', 128, 128, range(0, 128), range(0, 128), None),
    ('side mask [0,96)   rec mask [96,192)   main mask [64,256)', 96, 96, range(0, 96), range(96, 192), range(64, 256)),
    ('side mask even CUs rec mask odd CUs    main all', 128, 128, range(0, 256, 2), range(1, 256, 2), None),
    ('baseline again', 128, 128, None, None, None),
]
for name, sc, rc, smask, rmask, mmask in variants:
    ts = build(sc, rc)
    main = masked_stream(mmask) if mmask is not None else None
    if smask is not None:
        # HIP multiplexes streams onto four hardware queues: keep creating masked streams until one runs beside the others
        scratch = torch.zeros(256, device='cuda:0')
        mref = main or torch.cuda.current_stream()
        keep = []
        def pick(mask, beside):
            for _ in range(10):
                st = masked_stream(mask)
                keep.append(st)
                if all(ST.runs_beside(b, st, scratch) and ST.runs_beside(b, st, scratch) for b in beside):
                    return st
            return st
        ts.side = [pick(smask, [mref])]
        ts.rec_stream = pick(rmask, [mref, ts.side[0]])
    ms = time_steps(ts, main)
    ok = ''
    if smask is not None:
        scratch = torch.zeros(256, device='cuda:0')
        ok = '  (side beside rec: %s, side beside main: %s)' % (ST.runs_beside(ts.rec_stream, ts.side[0], scratch),
                                                                ST.runs_beside(main or torch.cuda.current_stream(), ts.side[0], scratch))
    print('%-62s %.3f ms/step%s' % (name, ms, ok), flush=True)
    del ts
