#!/usr/bin/env python3
"""rd_ram_mix alone at the bench geometry (B = 8, 400x400, bf16 slot output): mean time over 200 calls (HIP events) and the
algorithmic GB/s of SURVEY.md 8d (12*C*S^2 bytes per image: source + partner + output as fp32).
    python scripts/ram_bench.py [u8|f32] [S]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ram-dsir_amd')]
import torch
from ramdsir import ram as R
kind = sys.argv[1] if len(sys.argv) > 1 else 'u8'
S = int(sys.argv[2]) if len(sys.argv) > 2 else 400
B, dev = 8, 'cuda:0'
src = (torch.rand(B, S, S, 3, device=dev) * 255).round()
trg = (torch.rand(B, S, S, 3, device=dev) * 255).round()
if kind == 'u8':
    src, trg = src.to(torch.uint8), trg.to(torch.uint8)
lam = torch.full((B,), 0.4, device=dev)
m = R.RamMixer(B, S, S, torch.bfloat16, dev, 'fundus')
x = torch.zeros(2 * B, S, S, 8, dtype=torch.bfloat16, device=dev)
m.bind(src, trg, lam, x[:B], x[B:])
for _ in range(20):
    m.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 200
e0.record()
for _ in range(n):
    m.run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
print('rd_ram_mix %s S=%d B=%d: %.1f us per call, %.0f GB/s algorithmic (12*C*S^2 B/img = %.1f MB per call)'
      % (kind, S, B, us, 12 * 3 * S * S * B / us / 1e3, 12 * 3 * S * S * B / 1e6))
