#!/usr/bin/env python3
"""End-to-end throughput of train.py with real files: a synthetic Fundus tree of 800x800 RGB PNG ROIs
(the size of the reference's ROIs) + gray masks in the reference's list layout, then the drop-in CLI with the reference's
loader settings (train.py:558: batch [3,6,7] for target 0, num_workers=8 per domain loader, pin_memory, shuffle,
Resize(256) + RandomScaleCrop(256)).  Prints train.py's `train throughput` line; compare with bench.py's resident-input
number at --size 256.   python scripts/e2e_train_throughput.py [--n 48] [--iters 120] [--workers 8]"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'tests')]
import synth_data as SD

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=160)            # 160 / 3 = 53 iterations per epoch, like the real lists (train.py:210)
ap.add_argument('--iters', type=int, default=270)
ap.add_argument('--workers', type=int, default=8)
ap.add_argument('--dtype', default='bf16')
a = ap.parse_args()
with tempfile.TemporaryDirectory() as tmp:
    t0 = time.time()
    SD.make_fundus_tree(tmp, n_train=a.n, n_test=8, hw=(800, 800), vary=False)
    print('tree: 4 domains x %d train PNGs of ~800x800 in %.1f s' % (a.n, time.time() - t0), flush=True)
    cmd = [sys.executable, os.path.join(ROOT, 'ram-dsir_amd', 'train.py'), '--data_root', tmp, '--dataset', 'fundus', '--domain_idxs', '1,2,3',
           '--test_domain_idx', '0', '--ram', '--rec', '--is_out_domain', '--consistency', '--consistency_type', 'kd', '--save_path',
           os.path.join(tmp, 'out'), '--epochs', '1000', '--max_iters', str(a.iters), '--num_workers', str(a.workers), '--log_every', '50',
           '--dtype', a.dtype]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = r.stdout.decode()
    print('\n'.join(l for l in out.splitlines() if 'throughput' in l or 'epoch ' in l or 'Error' in l or 'error' in l)[-3000:])
    sys.exit(r.returncode)
