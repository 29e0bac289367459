"""Runs the synthetic-domain Dice proxy (tests/dice_proxy.py) and writes gpurun_out/r3/<tag>_dice_proxy.json (copied to profiles/):
held-out Dice of the oracle (CPU fp32), of the oracle under the bf16 rounding model, and of three HIP fp32 / three HIP bf16 runs
after N iterations from the same init and batch stream.
usage: dice_proxy_run.py [tag=r03] [iters=300] [reps=3]        (test infrastructure: imports oracle/)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'ram-dsir_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
import dice_proxy as DP
from oracle import unet as OU
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
train, test = DP.make_data()
stream = DP.batch_stream(train, n)
res = dict(iters=n, size=DP.S, batch=DP.BATCH, n_test=len(test), cup_ratio=DP.CUP_RATIO, cup_gain=DP.CUP_GAIN)


def record(name, fn, product=False):
    t0 = time.time()
    st, hist = fn()
    cup, disc = DP.evaluate_with_oracle(st, test)
    r = dict(cup_dice=round(100 * cup, 3), disc_dice=round(100 * disc, 3), avg_dice=round(50 * (cup + disc), 3),
             loss_first=round(hist[0], 4), loss_last=round(hist[-1], 4), seconds=round(time.time() - t0, 1))
    if product:
        pc, pd = DP.evaluate_with_product(st, test)
        r['avg_dice_product_eval'] = round(50 * (pc + pd), 3)
    res.setdefault(name, []).append(r)
    print(name, r, flush=True)


record('oracle_cpu_fp32', lambda: DP.train_oracle(stream))
with OU.rounding(torch.bfloat16):
    record('oracle_cpu_bf16_rounding_model', lambda: DP.train_oracle(stream))
for _ in range(reps):
    record('hip_fp32', lambda: DP.train_hip(stream, torch.float32), product=True)
for _ in range(reps):
    record('hip_bf16', lambda: DP.train_hip(stream, torch.bfloat16), product=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out', 'r3'), exist_ok=True)
with open(os.path.join(ROOT, 'gpurun_out', 'r3', '%s_dice_proxy.json' % tag), 'w') as f:
    json.dump(res, f, indent=1)
print(json.dumps(res))
