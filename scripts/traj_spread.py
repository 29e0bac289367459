"""Spread of the 20-step fixed-batch loss trajectory test (tests/test_gpu_fullsize.py::test_bf16_and_fp32_loss_trajectories_track_the_oracle):
mean / max |log(loss_hip / loss_oracle)| over repeated HIP runs, with the fused small-channel backward on and off.
(test infrastructure: imports oracle/)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'ram-dsir_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
import fullsize_util as FU
from oracle import step as OS, unet as OU
from ramdsir import tuning as T
cfg = FU.CONFIGS['T128']
src, trg, lam, mask = FU.synth(cfg)
states = FU.oracle_states(len(cfg['bs']))
img, frq = FU.oracle_ram(cfg, src, trg, lam)
enc, dec, rec = (OU.clone_state(s) for s in states)
opt = {m: OS.adam_state({k: sd[k] for k in OU.param_keys(sd)}) for m, sd in (('enc', enc), ('dec', dec), ('rec', rec))}
c = OS.StepConfig(dataset='fundus', batch_sizes=cfg['bs'], consistency='kd', lr=2e-3, total_iters=1000)
ref = []
for it in range(20):
    comps, _ = OS.train_step(enc, dec, rec, opt, img, frq, torch.from_numpy(mask), c, it)
    ref.append(comps['total'].item())
for fused in (True, False):
    T.DEFAULTS['fused_bwd'] = fused
    for dt in (torch.bfloat16, torch.float32):
        out = []
        for rep in range(4):
            ts, bank, got = FU.hip_step(cfg, states, src, trg, lam, mask, dt, total_iters=1000, nsteps=20)
            hist = [h[5] for h in got['hist']]
            logr = [abs(float(np.log(hist[i] / ref[i]))) for i in range(20)]
            out.append((round(float(np.mean(logr)), 3), round(max(logr), 3)))
            del ts, bank
        print('fused_bwd', fused, dt, '(mean, max) |log ratio| per run:', out, flush=True)
