"""Is the 300-step Dice-proxy training run bitwise repeatable?  N runs of tests/dice_proxy.py::train_hip in one process, the digest (sum of
every trained parameter in fp64) of each.  With the debug library, RD_* switches select kernels: an intermittent race shows as a digest
that differs between runs.  usage: determinism_probe.py [bf16|f32] [runs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd'), os.path.join(ROOT, 'tests')]
import torch
import dice_proxy as DP
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
train, test = DP.make_data()
stream = DP.batch_stream(train, 300)
digests = []
for r in range(n):
    states, hist = DP.train_hip(stream, dtype)
    d = 0.0
    for sd in states:
        for k, v in sd.items():
            if torch.is_tensor(v) and v.is_floating_point():
                d += float(v.double().sum())
    digests.append(d)
    print('run %d: digest %.12f  last loss %.6f' % (r, d, hist[-1]), flush=True)
print('REPEATABLE' if len(set(digests)) == 1 else 'DIFFERENT RUNS: %d distinct digests' % len(set(digests)))
