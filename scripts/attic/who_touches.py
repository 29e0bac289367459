"""Which launches of the step hold a pointer into a given plan tensor (seg.keep[i])?  Walks the ctypes descriptors of every op of seg_a / seg_b
and prints the field paths whose value lies inside the tensor.  usage: who_touches.py keep_index [side]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import step as S
import bench as Bn
ki = int(sys.argv[1]); Sz = int(sys.argv[2]) if len(sys.argv) > 2 else 64
bank, mods = S.make_bank('cuda:0', 3, 16, 2, 3)
Bn.init_weights(bank)
ts = S.TrainStep(bank, mods, torch.bfloat16, [2, 3, 3], Sz, Sz, dataset='fundus', consistency='kd', lr=2e-3, total_iters=1000, ram=None)
t = [x for x in ts.seg.keep if torch.is_tensor(x)][ki]
lo, hi = t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()
print('tensor', ki, tuple(t.shape), t.dtype, hex(lo), hex(hi))


def walk(obj, path, out, depth=0):
    if depth > 4:
        return
    if isinstance(obj, C.Structure):
        for name, _ in obj._fields_:
            walk(getattr(obj, name), path + '.' + name, out, depth + 1)
    elif isinstance(obj, C.Array):
        for i, v in enumerate(obj):
            walk(v, '%s[%d]' % (path, i), out, depth + 1)
    elif isinstance(obj, int):
        if lo <= obj < hi:
            out.append(path)
    elif obj is not None and hasattr(obj, 'value') and isinstance(getattr(obj, 'value', None), int):
        if lo <= obj.value < hi:
            out.append(path)


for seg in ('seg_a', 'seg_b'):
    for i, op in enumerate(getattr(ts, seg)):
        if op[0] is None:
            continue
        hits = []
        for j, a in enumerate(op[1]):
            tgt = a._obj if hasattr(a, '_obj') else a
            walk(tgt, 'arg%d' % j, hits)
        if hits:
            meta = op[2] if len(op) > 2 and isinstance(op[2], dict) else {}
            print('%s[%d] %s %s %s: %s' % (seg, i, getattr(op[0], '__name__', op[0]), meta.get('kernel'), meta.get('layer'), hits))
