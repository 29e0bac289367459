"""GPU: the spread of the HIP step's held-out Dice on the synthetic-domain proxy (tests/dice_proxy.py) over the SAME one-ulp-sized
perturbations of the initial weights as scripts/dice_proxy_oracle_spread.py uses for the oracle.  usage: dice_proxy_hip_spread.py [runs=8]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'ram-dsir_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import torch
import dice_proxy as DP
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
train, test = DP.make_data()
stream = DP.batch_stream(train, 300)
out = {}
for name, dtype in (('hip_bf16', torch.bfloat16), ('hip_fp32', torch.float32)):
    vals = []
    for seed in [None] + list(range(runs - 1)):
        st, hist = DP.train_hip(stream, dtype, perturb_seed=seed)
        c, d = DP.evaluate_with_oracle(st, test)
        vals.append(round(50.0 * (c + d), 3))
        print(name, seed, vals[-1], flush=True)
    out[name] = dict(avg_dice=vals, mean=round(float(np.mean(vals)), 3), sd=round(float(np.std(vals, ddof=1)), 3))
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'dice_proxy_hip_spread.json'), 'w'), indent=1)
