"""CPU only: the spread of the ORACLE's held-out Dice on the synthetic-domain proxy (tests/dice_proxy.py) over one-ulp-sized perturbations
of the initial weights, in fp32 and under the bf16 rounding model (oracle.unet.rounding) -- the distribution a single HIP run has to be
compared with.  usage: dice_proxy_oracle_spread.py [runs=6] [iters=300]      (test infrastructure: imports oracle/)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'ram-dsir_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import torch
import dice_proxy as DP
from oracle import unet as OU
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
train, test = DP.make_data()
stream = DP.batch_stream(train, n)
out = {'iters': n, 'perturbation': 'initial parameters x (1 + 1e-6 N(0,1)), seeds 0..%d; seed None = unperturbed' % (runs - 1)}
for name, ctx in (('oracle_bf16_rounding_model', lambda: OU.rounding(torch.bfloat16)), ('oracle_fp32', None)):
    vals = []
    for seed in [None] + list(range(runs - 1)):
        t0 = time.time()
        if ctx is not None:
            with ctx():
                st, hist = DP.train_oracle(stream, perturb_seed=seed)
        else:
            st, hist = DP.train_oracle(stream, perturb_seed=seed)
        c, d = DP.evaluate_with_oracle(st, test)
        vals.append(round(50.0 * (c + d), 3))
        print(name, seed, vals[-1], '%.0f s' % (time.time() - t0), flush=True)
    out[name] = dict(avg_dice=vals, mean=round(float(np.mean(vals)), 3), sd=round(float(np.std(vals, ddof=1)), 3))
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, 'profiles'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'profiles', 'r05_dice_proxy_oracle_spread.json'), 'w'), indent=1)
