#!/usr/bin/env python3
"""bench.py -- images/sec of the full RAM-DSIR training step (RAM FFT + seg on img and img_freq +
consistency + per-domain rec + backward + Adam) on N MI355X of one node.

  python bench.py --gpus 1 --steps 100 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): Fundus target 0 --ram --rec --consistency kd, bf16 storage / fp32
accumulate, batch 8 = [2,3,3] per GPU at 400x400x3, synthetic images (no dataset is available offline),
random-init weights.  Weak scaling: every rank runs its own batch of 8; gradients are averaged with RCCL.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'ram-dsir_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

# Kernel families of the step (one entry = launches of the list selected by `match` on ramdsir/engine.py's meta, with the kernel
# symbols rocprofv3 reports for them; `count` = the symbols that count as ONE launch where a launch is two kernels):
#   bwd_fused  dgrad + weight gradient of the <= 32-channel 3x3 convs in one launch (csrc/conv_fused.hip), HBM-bound
#   conv64     3x3 convs on 64-wide output-channel tiles (conv_ws_kernel + conv_pf_kernel<bf16,9,2,*> + conv_kernel<bf16,9,2> + conv_pp_kernel), MFMA-bound
#   wgrad      the stand-alone weight gradients (>= 64-channel layers -- wgrad_sym_kernel for whole 128-channel gradient blocks, else wgrad_ws_kernel --,
#              1x1 convs, the first conv): MFMA kernel + split reduce
#   conv_small forward (and the few unfused gradient) launches of the <= 32-channel convs (conv_small_fwd_kernel / conv_small_kernel), HBM-bound
#   ram        Random Amplitude Mixup: rd_ram_mix = kept row bins (matrix-core DFT for uint8 images, else row FFT) + column FFT / window mix /
#              inverse column FFT + inverse row FFT with the output epilogue (3 kernels), HBM-bound
# The HEADLINE `roofline` is not a constant: bench.py times the step with each of the three families that hold the most kernel time
# left out (scripts/ablate_step.py's measurement, inside this run) and headlines the one whose absence shortens the step most
# (`dominant_by_step_cost`; costs within 10 % of the largest are a tie, which goes to the family with the most kernel time);
# `dominant_by_kernel_time` is reported beside it.  The other families follow as roofline_<name>.
FAMILIES = {
    'bwd_fused': dict(match=lambda m: m.get('kernel') == 'conv_small_bwd_fused', symbols=('conv_small_bwd_fused_kernel',), count=None),
    'conv64': dict(match=lambda m: m.get('kernel') == 'conv_kernel<bf16,9,2>',
                   symbols=('conv_ws_kernel', 'conv_pf_kernelIDF16bLi9ELi2E', '11conv_kernelIDF16bLi9ELi2E', 'conv_pp_kernel'), count=None),
    'wgrad': dict(match=lambda m: m.get('kernel') == 'wgrad',
                  symbols=('wgrad_sym_kernel', 'wgrad_ws_kernel', 'wgrad_tr_kernel', 'wgrad_c16_tr_kernel', 'wgrad_kernel', 'wgrad_reduce_kernel'),
                  count=('wgrad_sym_kernel', 'wgrad_ws_kernel', 'wgrad_tr_kernel', 'wgrad_c16_tr_kernel', 'wgrad_kernel')),
    'conv_small': dict(match=lambda m: str(m.get('kernel', '')).startswith('conv_small_kernel'),
                       symbols=('conv_small_fwd_kernel', 'conv_small_kernel'), count=None),
    'ram': dict(match=lambda m: m.get('kernel') == 'ram', symbols=('ram_row_dft_kernel', 'ram_row_fwd_kernel', 'ram_col_mix_kernel', 'ram_row_inv_kernel'),
                count=('ram_col_mix_kernel',)),
}
PMC_JSON = os.path.join(ROOT, 'profiles', 'dominant_kernel_pmc.json')
SQ_JSON = os.path.join(ROOT, 'profiles', 'r04_mfma_busy.json')
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
HBM_COPY_GBS = 5300.0                    # what a plain torch copy (read + write) sustains on this part: profiles/r03_hbm_ceiling.txt
MFMA_PEAK_TFLOPS = 2500.0                # MI355X_MICROARCH.md: dense bf16 MFMA ~2.5 PF (no sparsity)


# The workloads BASELINE.json lists: dataset (loss kind, RAM input range), per-domain batch split, side, classes.
#   C2   configs[1], the metric's configuration: Fundus target 0, 3 source domains x [2,3,3] at 400 x 400 (train.py:35-37 scaled to 8)
#   C3   configs[2]: Prostate leave-one-out, 5 source domains x 2 slices at 384 x 384, softmax / CE / dice_multi (train.py:39-45,363-465)
#   C5   configs[4]: synthetic 512 x 512 stream, 4 source domains x 2 (the per-GPU step of the 8-GPU roofline run)
#   F256 the reference's own training shape: Fundus 16 = [3,6,7] at 256 x 256 (train.py:35,541)
CONFIGS = {
    'C2': dict(dataset='fundus', bs=[2, 3, 3], size=400, num_classes=2, consistency='kd', lr=2e-3,
               workload='Fundus target0 --ram --rec --consistency kd, batch 8=[2,3,3] per GPU, %dx%dx3'),
    # lr 1e-3 is the reference's Prostate default (train.py:619).  --consistency_type mse, not kd: on white-noise images the softmax of
    # the freshly initialised network reaches EXACT zeros within two steps (logit differences > 100), and the reference's own KD formula
    # (train.py:85-88: KLDivLoss of p.log()) is inf / NaN there -- in the oracle as in the HIP loss kernel; the launches are the same
    'C3': dict(dataset='prostate', bs=[2, 2, 2, 2, 2], size=384, num_classes=2, consistency='mse', lr=1e-3,
               workload='Prostate leave-one-out --ram --rec --consistency --consistency_type mse, batch 10=[2,2,2,2,2] per GPU (5 source domains), %dx%dx3 (2.5-D slices)'),
    'C5': dict(dataset='fundus', bs=[2, 2, 2, 2], size=512, num_classes=2, consistency='kd', lr=2e-3,
               workload='Synthetic 4-domain stream --ram --rec --consistency kd, batch 8=[2,2,2,2] per GPU, %dx%dx3'),
    'F256': dict(dataset='fundus', bs=[3, 6, 7], size=256, num_classes=2, consistency='kd', lr=2e-3,
                 workload='Fundus, the reference\'s own training shape (train.py:35,541) --ram --rec --consistency kd, batch 16=[3,6,7] per GPU, %dx%dx3'),
}


def synth_inputs(B, S, rank, device, dataset='fundus'):
    """SURVEY.md 8(d).  fundus: U[0,255] uint8 source and partner images (what a decoded PNG is), masks = two concentric random discs
    (cup inside disc) as the 2-channel multilabel float mask; prostate: float32 U[-1,1] slices (what the .npy files hold,
    prostate.py:177-188), int64 label map = one random disc.  lambda from random.Random(1337) in both."""
    rng = np.random.RandomState(1337 + rank)
    if dataset == 'fundus':
        src = np.round(rng.uniform(0, 255, (B, S, S, 3))).astype(np.uint8)
        trg = np.round(rng.uniform(0, 255, (B, S, S, 3))).astype(np.uint8)
    else:
        src = rng.uniform(-1, 1, (B, S, S, 3)).astype(np.float32)
        trg = rng.uniform(-1, 1, (B, S, S, 3)).astype(np.float32)
    pr = random.Random(1337 + rank)
    lam = np.array([pr.randint(1, 10) / 10 for _ in range(B)], np.float32)
    yy, xx = np.mgrid[0:S, 0:S]
    mask = np.zeros((B, 2, S, S), np.float32)
    for i in range(B):
        cy, cx = rng.uniform(0.35 * S, 0.65 * S, 2)
        r_disc = rng.uniform(0.15 * S, 0.3 * S)
        r_cup = r_disc * rng.uniform(0.3, 0.7)
        d2 = (yy - cy) ** 2 + (xx - cx) ** 2
        mask[i, 1] = d2 <= r_disc ** 2
        mask[i, 0] = d2 <= r_cup ** 2
    if dataset != 'fundus':
        mask = mask[:, 1].astype(np.int64)
    t = lambda a: torch.from_numpy(a).to(device)
    return t(src), t(trg), t(lam), t(mask), (src, trg, lam, mask)


def init_weights(bank):
    """Reference init (unet.py:257-262): kaiming_normal(fan_out) conv weights, default conv bias, BN w=1 b=0."""
    g = torch.Generator(device='cpu').manual_seed(1337)
    for (m, k), (off, shape) in bank.index.items():
        v = bank.p(m, k)
        if len(shape) == 4:
            v.copy_((torch.randn(shape, generator=g) * (2.0 / (shape[0] * shape[2] * shape[3])) ** 0.5).to(v.device))
        elif '.bn' in k and k.endswith('weight'):
            v.fill_(1.0)
        elif '.bn' in k:
            v.zero_()
        else:
            wshape = bank.index[(m, k[:-len('bias')] + 'weight')][1]
            bound = 1.0 / (wshape[1] * wshape[2] * wshape[3]) ** 0.5
            v.copy_(((torch.rand(shape, generator=g) * 2 - 1) * bound).to(v.device))


_PMC_LIVE = None            # filled by collect_live_pmc(): the same structure as the committed json, measured by THIS run


def _pmc_json():
    if _PMC_LIVE is not None:
        return _PMC_LIVE
    if not os.path.exists(PMC_JSON):
        return {}
    with open(PMC_JSON) as f:
        return json.load(f)


def _traffic_source():
    if _PMC_LIVE is not None:
        return _PMC_LIVE['collected']
    return 'profiles/dominant_kernel_pmc.json (%s)' % _pmc_json().get('collected', 'rocprofv3 PMC, offline')


def pmc_aggregate(pmc_rows, B, dtype, size):
    """rocprofv3 counter rows (Kernel_Name, Counter_Name in FETCH_SIZE / WRITE_SIZE, Counter_Value in KB; the two counters from
    separate passes) -> HBM bytes per launch of every kernel family and per step, corrected as MI355X_MICROARCH.md prescribes:
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  One launch of a family = the kernels its 'count' symbols name (a weight-gradient
    launch = MFMA kernel + its reduce).  Shared by scripts/pmc_traffic.py (the committed summary) and collect_live_pmc()."""
    out = {'correction': 'HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: FETCH_SIZE reports half of 16-B/lane '
                         'streaming reads on gfx950; counters in KB); FETCH_SIZE and WRITE_SIZE collected in separate passes'}
    for fam, spec in FAMILIES.items():
        member = lambda n: any(f in n for f in spec['symbols'])
        counted = lambda n: any(f in n for f in (spec['count'] or spec['symbols']))
        kb = {'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0}
        nl = {'FETCH_SIZE': 0, 'WRITE_SIZE': 0}
        for r in pmc_rows:
            if member(r['Kernel_Name']):
                kb[r['Counter_Name']] += float(r['Counter_Value'])
                if counted(r['Kernel_Name']):
                    nl[r['Counter_Name']] += 1
        if min(nl.values()) == 0:
            continue
        fk, wk = kb['FETCH_SIZE'] / nl['FETCH_SIZE'], kb['WRITE_SIZE'] / nl['WRITE_SIZE']
        out[fam] = {'family': fam, 'kernels': ' + '.join(spec['symbols']), 'launches_profiled': nl['FETCH_SIZE'],
                    'fetch_size_kb_per_launch': fk, 'write_size_kb_per_launch': wk, 'traffic_bytes_per_launch': (2 * fk + wk) * 1024}
    # every kernel of a step: all dispatches of the library's kernels (torch's allocation fills excluded: they run once, before the
    # first step) divided by the number of steps the profiled command ran (= dispatches of the once-per-step Adam kernel)
    ours = lambda n: not n.startswith('_ZN2at') and 'at::native' not in n
    nsteps = {c: sum(1 for r in pmc_rows if r['Counter_Name'] == c and 'adam_update_kernel' in r['Kernel_Name']) for c in ('FETCH_SIZE', 'WRITE_SIZE')}
    if min(nsteps.values()) > 0:
        tot_kb = {c: sum(float(r['Counter_Value']) for r in pmc_rows if r['Counter_Name'] == c and ours(r['Kernel_Name'])) for c in nsteps}
        fk, wk = tot_kb['FETCH_SIZE'] / nsteps['FETCH_SIZE'], tot_kb['WRITE_SIZE'] / nsteps['WRITE_SIZE']
        per_kernel = {}
        for r in pmc_rows:
            if ours(r['Kernel_Name']):
                k = r['Kernel_Name'].split('(')[0][:60]
                per_kernel.setdefault(k, {'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0})[r['Counter_Name']] += float(r['Counter_Value'])
        top = sorted(per_kernel.items(), key=lambda kv: -(2 * kv[1]['FETCH_SIZE'] / nsteps['FETCH_SIZE'] + kv[1]['WRITE_SIZE'] / nsteps['WRITE_SIZE']))[:12]
        out['step'] = {'steps_profiled': nsteps['FETCH_SIZE'], 'fetch_size_kb_per_step': fk, 'write_size_kb_per_step': wk,
                       'traffic_bytes_per_step': (2 * fk + wk) * 1024, 'size': size, 'dtype': dtype, 'batch': B,
                       'top_kernels_mb_per_step': {k: round((2 * v['FETCH_SIZE'] / nsteps['FETCH_SIZE'] + v['WRITE_SIZE'] / nsteps['WRITE_SIZE']) / 1024, 1) for k, v in top}}
    return out


def collect_live_pmc(B, dtype, size, config='C2', timeout=240):
    """HBM traffic measured BY THIS RUN: the same command (3 steps, no baselines) as a child process under `rocprofv3 -i scripts/pmc_hbm.txt
    --kernel-trace` (FETCH_SIZE and WRITE_SIZE in separate passes, kernel trace only: the guide's recipe), its counter files
    aggregated by pmc_aggregate().  Returns True when the figures of the line come from it; on any failure (no rocprofv3, a
    time-out, unreadable output) the line falls back to the committed profiles/dominant_kernel_pmc.json and says so."""
    global _PMC_LIVE
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which('rocprofv3')
    if exe is None or os.environ.get('RD_BENCH_CHILD') == '1':
        return False
    # not from inside a profiler: when this process itself runs under rocprofv3 (the committed --stats / --pmc collections) its
    # environment carries the tool library, and a nested profiler would inherit it
    if _under_profiler():
        return False
    outdir = tempfile.mkdtemp(prefix='rd_pmc_', dir='/tmp')
    try:
        cmd = [exe, '-i', os.path.join(ROOT, 'scripts', 'pmc_hbm.txt'), '--kernel-trace', '-M', '--output-format', 'csv', '-d', outdir, '-o', 'p',
               '--', sys.executable, os.path.abspath(__file__), '--steps', '3', '--warmup', '1', '--config', config, '--size', str(size), '--dtype', dtype,
               '--no-cpu-baseline', '--no-fp32-leg', '--no-ablation', '--no-live-pmc', '--no-saturation']
        env = dict(os.environ, TMPDIR='/tmp', RD_BENCH_CHILD='1')
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'RD_FORCE_DDP'):
            env.pop(k, None)
        r = subprocess.run(cmd, cwd='/tmp', env=env, timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        rows = []
        for f in glob.glob(os.path.join(outdir, '**', '*counter_collection.csv'), recursive=True):
            with open(f) as fh:
                rows += [x for x in csv.DictReader(fh) if x['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE')]
        agg = pmc_aggregate(rows, B, dtype, size)
        if r.returncode != 0 or 'step' not in agg:
            return False
        agg['collected'] = ('live: rocprofv3 -i scripts/pmc_hbm.txt --kernel-trace around a %d-step child run of this command '
                            '(FETCH_SIZE / WRITE_SIZE in separate passes)' % agg['step']['steps_profiled'])
        _PMC_LIVE = agg
        return True
    except Exception:
        return False
    finally:
        shutil.rmtree(outdir, ignore_errors=True)


def kernel_roofline(ts, fam, eager=True):
    """One more step, launched exactly like the timed ones (eager: weight-gradient kernels on the side stream and the
    restoration-decoder branch on its own stream, so the timed launches see the same contention), with HIP events
    recorded on the stream each launch goes to around every launch of the kernel family; the algorithmic
    bytes / flops of each launch come from its descriptor (engine.Plan._conv_meta, ram.RamMixer.op)."""
    match = FAMILIES[fam]['match']
    main = torch.cuda.current_stream()
    ts.zero()
    evs, acc = [], dict(nbytes=0, flops=0)

    def wrap(op, stream, launch):
        meta = op[2] if len(op) > 2 else None
        if not meta or not match(meta):
            return launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        launch()
        e1.record(stream)
        evs.append((e0, e1))
        acc['nbytes'] += meta['bytes']
        acc['flops'] += meta['flops']
        acc['built'] = acc.get('built', 0) + meta.get('bytes_as_built', 0)

    lanes = ts.lanes() if eager else {}
    ts.run_segment(ts.seg_a + ts.seg_b, main, lanes, wrap)
    ts.run_segment(ts.seg_c, main, lanes, wrap)
    nbytes, flops = acc['nbytes'], acc['flops']
    torch.cuda.synchronize()
    total_ms = sum(a.elapsed_time(b) for a, b in evs)
    n = len(evs)
    gbs = nbytes / (total_ms * 1e-3) / 1e9
    tfs = flops / (total_ms * 1e-3) / 1e12
    # which roof binds: arithmetic intensity of the launches against the ridge point of the chip
    intensity, ridge = flops / nbytes, MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
    # HBM bytes per launch from the PMC counters of the same kernels: NOT measured in this run (PMC collection needs
    # rocprofv3 around the process) but read from the committed summary of `rocprofv3 -i scripts/pmc_hbm.txt` (separate
    # FETCH_SIZE / WRITE_SIZE passes) on this command, corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 for
    # 16-B/lane streaming reads, KB -> B); `traffic_source` names the file, profiles/README.md the procedure
    traffic, traffic_source = None, None
    tj = _pmc_json().get(fam, {})
    if tj.get('family') == fam and 'traffic_bytes_per_launch' in tj:
        traffic = int(tj['traffic_bytes_per_launch'])
        traffic_source = _traffic_source()
    if intensity >= ridge:
        out = dict(bound='mfma', achieved=round(tfs, 1), peak=MFMA_PEAK_TFLOPS, unit='TFLOP/s', frac=round(tfs / MFMA_PEAK_TFLOPS, 4))
    else:
        out = dict(bound='hbm', achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(gbs / HBM_PEAK_GBS, 4),
                   frac_of_copy_rate=round(gbs / HBM_COPY_GBS, 4))    # beside the spec: the rate streaming kernels reach in practice
    out.update(traffic=traffic, traffic_source=traffic_source, family=fam, kernels=' + '.join(FAMILIES[fam]['symbols']), launches_per_step=n,
               avg_launch_us=round(total_ms * 1e3 / n, 1), step_kernel_us=round(total_ms * 1e3, 1),
               avg_algorithmic_bytes=int(nbytes / n), avg_flops=int(flops / n), flop_per_byte=round(intensity, 1),
               achieved_gbs=round(gbs, 1), achieved_tflops=round(tfs, 1))
    if traffic:
        out['traffic_over_algorithmic'] = round(traffic / (nbytes / n), 3)
    if acc.get('built'):
        # the same launches priced on the bytes of the path AS BUILT (uint8 pixels in, fp32 spectra of the kept bins between the passes,
        # one 16-byte slot per pixel and output tensor out) beside SURVEY.md 8(d)'s fp32-in / fp32-out convention above
        built = acc['built'] / n
        out.update(avg_bytes_as_built=int(built), achieved_gbs_as_built=round(built * n / (total_ms * 1e-3) / 1e9, 1),
                   frac_as_built=round(built * n / (total_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
        if traffic:
            out['traffic_over_as_built'] = round(traffic / built, 3)
    return out


_SQ_LIVE = None             # filled by collect_live_sq()


def sq_aggregate(acc):
    """SQ counter means per (kernel, grid, workgroup size) -> per family: mfma_busy = sum SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel
    cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs); mfma_busy_occupied divides by the SIMDs of the
    CUs the launch can occupy; lds_conflict_ratio = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.  Shared with scripts/mfma_busy.py."""
    dem = {'conv_pf_kernelIDF16bLi9ELi2E': 'conv_pf_kernel<__bf16, 9, 2', '11conv_kernelIDF16bLi9ELi2E': 'conv_kernel<__bf16, 9, 2'}
    mean = lambda v: sum(v) / len(v)
    out = {}
    for fam, spec in FAMILIES.items():
        busy = cyc = occ = n = conf = ldsact = 0.0
        for (name, grid, wgs), c in acc.items():
            if not any(dem.get(sym, sym) in name for sym in spec['symbols']):
                continue
            if 'SQ_VALU_MFMA_BUSY_CYCLES' not in c or 'GRBM_GUI_ACTIVE' not in c:
                continue
            k = len(c['GRBM_GUI_ACTIVE'])
            kc = mean(c['GRBM_GUI_ACTIVE']) / 8.0
            cus = min(int(grid) // max(int(wgs), 1), 256)
            busy += k * mean(c['SQ_VALU_MFMA_BUSY_CYCLES'])
            cyc += k * 1024.0 * kc
            occ += k * 4.0 * cus * kc
            n += k
            if 'SQ_LDS_BANK_CONFLICT' in c and 'SQ_LDS_IDX_ACTIVE' in c:
                conf += k * mean(c['SQ_LDS_BANK_CONFLICT'])
                ldsact += k * mean(c['SQ_LDS_IDX_ACTIVE'])
        if n and cyc:
            out[fam] = dict(mfma_busy=round(busy / cyc, 4), mfma_busy_occupied=round(busy / max(occ, 1.0), 4), dispatches=int(n))
            if ldsact:
                out[fam]['lds_conflict_ratio'] = round(conf / ldsact, 4)
    return out


def read_counter_dir(d):
    import collections, csv, glob
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                acc[(r['Kernel_Name'], r['Grid_Size'], r['Workgroup_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return acc


def _under_profiler():
    return any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ) or 'rocprof' in os.environ.get('LD_PRELOAD', '')


def collect_live_sq(dtype, size, timeout=240):
    """Matrix-pipe busy fraction and LDS conflict ratio per family measured by THIS run: scripts/pmc_step.py (one eager step on one
    stream) as a child under `rocprofv3 -i scripts/pmc_sq.txt --kernel-trace`.  False (-> the committed profiles/r04_mfma_busy.json)
    when it cannot run."""
    global _SQ_LIVE
    import shutil, subprocess, tempfile
    exe = shutil.which('rocprofv3')
    if exe is None or os.environ.get('RD_BENCH_CHILD') == '1' or _under_profiler():
        return False
    outdir = tempfile.mkdtemp(prefix='rd_sq_', dir='/tmp')
    try:
        cmd = [exe, '-i', os.path.join(ROOT, 'scripts', 'pmc_sq.txt'), '--kernel-trace', '--output-format', 'csv', '-d', outdir, '-o', 'p',
               '--', sys.executable, os.path.join(ROOT, 'scripts', 'pmc_step.py'), dtype, str(size), '2']
        env = dict(os.environ, TMPDIR='/tmp', RD_BENCH_CHILD='1')
        for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'RD_FORCE_DDP'):
            env.pop(k, None)
        r = subprocess.run(cmd, cwd='/tmp', env=env, timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        agg = sq_aggregate(read_counter_dir(outdir))
        if r.returncode != 0 or not agg:
            return False
        agg['collected'] = 'live: rocprofv3 -i scripts/pmc_sq.txt --kernel-trace around scripts/pmc_step.py %s %d 2 (one eager step, one stream)' % (dtype, size)
        _SQ_LIVE = agg
        return True
    except Exception:
        return False
    finally:
        shutil.rmtree(outdir, ignore_errors=True)


def mfma_busy(fam):
    """MFMA utilisation of a kernel family: SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles), from collect_live_sq() when it ran, else
    from the committed SQ-counter summary (scripts/collect_sq.sh -> scripts/mfma_busy.py -> profiles/r04_mfma_busy.json)."""
    if _SQ_LIVE is not None:
        j, src = _SQ_LIVE, _SQ_LIVE['collected']
    else:
        if not os.path.exists(SQ_JSON):
            return None
        with open(SQ_JSON) as f:
            j = json.load(f)
        src = '%s (%s)' % (os.path.relpath(SQ_JSON, ROOT), j.get('collected', 'rocprofv3 SQ counters, offline'))
    if fam not in j:
        return None
    out = dict(mfma_busy=j[fam]['mfma_busy'], mfma_busy_source=src)
    if 'lds_conflict_ratio' in j[fam]:
        out['lds_conflict_ratio'] = j[fam]['lds_conflict_ratio']        # SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
    return out


def time_steps(step, n, warm=3):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def step_cost_by_ablation(ts, fams, steps=20):
    """What a family costs THE STEP (scripts/ablate_step.py inside the bench run): the step is timed with that family's launches left
    out of the launch list (three streams, as the timed region runs it).  The skipped kernels' outputs keep the values of the last
    complete step, so the remaining kernels run on realistic data -- the ablated steps compute nothing meaningful, only their
    duration is used; this therefore runs AFTER everything that reads the state.  Returns {family: ms the step gets shorter}."""
    full = (list(ts.seg_a), list(ts.seg_b), list(ts.seg_c))
    rounds = 3                      # the three largest families cost within 0.06 ms of each other: median of three alternating rounds
    t = {fam: [] for fam in fams}
    bases = []
    try:
        for _ in range(rounds):
            bases.append(time_steps(ts.run_eager, steps))
            for fam in fams:
                match = FAMILIES[fam]['match']
                keep = lambda op: op[0] is None or len(op) < 3 or not match(op[2])
                ts.seg_a, ts.seg_b, ts.seg_c = ([op for op in seg if keep(op)] for seg in full)
                t[fam].append(time_steps(ts.run_eager, steps))
                ts.seg_a, ts.seg_b, ts.seg_c = full
        bases.append(time_steps(ts.run_eager, steps))
    finally:
        ts.seg_a, ts.seg_b, ts.seg_c = full
    med = lambda v: sorted(v)[len(v) // 2]
    base = med(bases)
    return {fam: round(base - med(t[fam]), 3) for fam in fams}, round(min(bases), 3), round(max(bases), 3)


def cpu_baseline(host_inputs, bs, dataset='fundus', num_classes=2, consistency='kd', lr=2e-3):
    """The oracle (torch CPU restatement of train.py:225-296 + numpy RAM of fundus.py:13-61) timed on this box's host
    cores on the same workload (8 images at 400x400 per step): 1 warm-up step + 3 timed steps, the numpy
    RAM per image on ONE core as a DataLoader worker runs it.  `value` = images/s of step + RAM / 8 workers (the
    reference's num_workers=8, train.py:558, assuming perfect overlap across the workers); `step_only` and
    `step_plus_ram_serial` are reported beside it."""
    from oracle import ram as OR, step as OS, unet as OU
    src, trg, lam, mask = host_inputs
    # torch's CPU conv kernels stop scaling (and then regress) well before the 100+ cores of a GPU host: 16 threads is
    # the fastest setting measured for this step (RD_CPU_THREADS overrides); `cores` reports the threads actually used
    cores = int(os.environ.get('RD_CPU_THREADS', '0')) or min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    B = src.shape[0]
    t0 = time.time()
    ram_fn = OR.ram_fundus if dataset == 'fundus' else OR.ram_prostate
    pairs = [ram_fn(src[i].astype(np.float32), trg[i].astype(np.float32), float(lam[i]), dtype=np.float32) for i in range(B)]
    t_ram = time.time() - t0                   # B images, one core
    img = torch.from_numpy(np.stack([p[0] for p in pairs]))
    frq = torch.from_numpy(np.stack([p[1] for p in pairs]))
    enc, dec, rec = OU.encoder_state(seed=1), OU.decoder_state(num_classes=num_classes, seed=2), OU.rec_decoder_state(num_classes=3, num_domains=len(bs), seed=3)
    opt = dict(enc=OS.adam_state({k: enc[k] for k in OU.param_keys(enc)}), dec=OS.adam_state({k: dec[k] for k in OU.param_keys(dec)}),
               rec=OS.adam_state({k: rec[k] for k in OU.param_keys(rec)}))
    cfg = OS.StepConfig(dataset=dataset, batch_sizes=bs, consistency=consistency, num_classes=num_classes, lr=lr)
    times = []
    for it in range(4):                        # 1 warm-up + 3 timed
        t0 = time.time()
        OS.train_step(enc, dec, rec, opt, img, frq, torch.from_numpy(mask), cfg, it)
        times.append(time.time() - t0)
    t_step = float(np.median(times[1:]))
    S_ = src.shape[1]
    return dict(value=round(B / (t_step + t_ram / 8), 3), unit='images/s', cores=cores, host_cores=os.cpu_count(), kind='port',
                step_only=round(B / t_step, 3), step_plus_ram_serial=round(B / (t_step + t_ram), 3),
                sample='1 warm-up + 3 timed steps of the same workload (%d images %dx%d each): torch-CPU step median %.2f s on %d threads of the '
                       'host\'s %d (warm-up %.2f s); numpy RAM %.2f s per batch of %d on 1 core; value = step + RAM/8 workers'
                       % (B, S_, S_, t_step, cores, os.cpu_count() or 0, times[0], t_ram, B))


def fp32_leg(bank_init, bs, Sz, dev, src, trg, lam, mask, steps=6, warmup=2, dataset='fundus', num_classes=2, consistency='kd', lr=2e-3):
    """The same step in fp32 storage -- the reference's own precision and the parity path of the tests."""
    from ramdsir import step as S
    bank, mods = S.make_bank(dev, 3, 16, num_classes, len(bs))
    bank.params.copy_(bank_init)
    ts = S.TrainStep(bank, mods, torch.float32, bs, Sz, Sz, dataset=dataset, consistency=consistency, lambda_rec=0.1, lr=lr,
                     total_iters=21200, num_classes=num_classes, ram='u8' if dataset == 'fundus' else True)
    ts.wpack.refresh()
    ts.load_raw(src, trg, lam)
    ts.load_target(mask)
    for _ in range(warmup):
        ts.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ts.step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out = dict(dtype='f32', images_per_s=round(sum(bs) * steps / el, 1), ms_per_step=round(1e3 * el / steps, 3), steps=steps,
               final_loss=round(ts.loss_dict()['loss'], 4))
    del ts, bank
    torch.cuda.empty_cache()
    return out


def bucket_allreduce_us(runner, world, reps=10):
    """Mean time of each gradient bucket's all-reduce alone (HIP events on the stream the step launches it from), in the
    order the step issues them (decoders, encoder levels 3-5, encoder levels 1-2), with the payload sizes."""
    import torch.distributed as dist
    comm, out = runner.comm, []
    for i in (2, 1, 0):
        t = runner.buckets.views[i]
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(comm):
            e0.record(comm)
            for _ in range(reps):
                w = runner.buckets.reduce(i, async_op=True)
                if w is not None:
                    w.wait()
            e1.record(comm)
        torch.cuda.synchronize()
        out.append(dict(bucket=i, bytes=int(t.numel() * t.element_size()), allreduce_us=round(e0.elapsed_time(e1) * 1e3 / reps, 1)))
    return out


def whole_step_roofline(B, Sz, ms_per_step, dtype):
    """SURVEY.md 8(d) conventions for the WHOLE step: algorithmic bytes = every conv reads its logical input once and
    writes its output once, everything else fused (170.3 M elements forward per image at 400x400, x3 for forward +
    backward, x (S/400)^2) + the RAM bytes 12*C*S^2; algorithmic flops = conv MACs x 2, forward + dgrad + wgrad
    (170.9 GFLOP per image at 400x400)."""
    esz = 2 if dtype == 'bf16' else 4
    sc = (Sz / 400.0) ** 2
    nbytes = B * (3 * 170.3e6 * esz * sc + 12 * 3 * Sz * Sz)
    flops = B * 170.9e9 * sc
    t = ms_per_step * 1e-3
    out = dict(algorithmic_gbs=round(nbytes / t / 1e9, 1), hbm_frac=round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4),
               algorithmic_tflops=round(flops / t / 1e12, 1), mfma_frac=round(flops / t / 1e12 / MFMA_PEAK_TFLOPS, 4),
               bytes_per_image=int(nbytes / B), flops_per_image=int(flops / B), algorithmic_bytes_per_step=int(nbytes))
    # HBM bytes of EVERY kernel of one step from the PMC passes (scripts/pmc_traffic.py: sum over all dispatches of the profiled
    # bench run / steps profiled), only for the workload the passes were collected on
    st = _pmc_json().get('step', {})
    if st.get('size') == Sz and st.get('dtype') == dtype and st.get('batch') == B and 'traffic_bytes_per_step' in st:
        out.update(traffic=int(st['traffic_bytes_per_step']), traffic_over_algorithmic=round(st['traffic_bytes_per_step'] / nbytes, 3),
                   traffic_gbs=round(st['traffic_bytes_per_step'] / t / 1e9, 1),
                   traffic_source=_traffic_source())
    else:
        out.update(traffic=None)
    return out


# reference box for `value_normalised`: the median of what the boxes of round 6 measured on the two micro-kernels (profiles/r06_box_probe.txt:
# copy 4 873-4 908 GB/s, MFMA loop 2 416-2 430 TFLOP/s = 0.97 of the dense bf16 peak)
BOX_REF = {'copy_gbs': 4900.0, 'mfma_tflops': 2425.0}


def shader_clock(dev, step):
    """Average shader clock (GHz) over 20 complete steps and over each box micro-kernel: two stamps of rd_box_probe(2) -- the
    shader-clock counter and the constant 100 MHz counter -- in stream order around the work, after the timed region.  A single
    launch on an idle device runs at 1.9-2.4 GHz depending on its instruction mix (docs/experiments.md, round 6); this says what the
    clock is in the steady state the bench times, i.e. whether the peaks the roofline quotes (at the 2.4 GHz boost clock) apply."""
    from ramdsir import _lib as L
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    stamps = torch.zeros(2, 2048, 2, dtype=torch.int64, device=dev)   # [before / after][unit][shader clock, 100 MHz clock]
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    n = 1 << 28
    a = torch.empty(n, dtype=torch.uint8, device=dev).fill_(1)
    b = torch.empty(n, dtype=torch.uint8, device=dev)

    def ghz(work):
        work()                                                        # warm
        torch.cuda.synchronize()
        assert lib.rd_box_probe(2, stamps[0].data_ptr(), None, 0, st) == 0
        work()
        assert lib.rd_box_probe(2, stamps[1].data_ptr(), None, 0, st) == 0
        torch.cuda.synchronize()
        both = (stamps[0, :, 1] != 0) & (stamps[1, :, 1] != 0)        # units stamped both times (counters of different XCDs are not aligned)
        d = (stamps[1] - stamps[0])[both].double()
        stamps.zero_()
        return round(float((0.1 * d[:, 0] / d[:, 1].clamp_min(1)).median()), 3) if int(both.sum()) else None

    def steps():
        for _ in range(20):
            step()

    def mfma():
        for _ in range(2):
            assert lib.rd_box_probe(1, sink.data_ptr(), None, 40000, st) == 0

    def copy():
        for _ in range(40):
            assert lib.rd_box_probe(0, a.data_ptr(), b.data_ptr(), n, st) == 0
    out = dict(step=ghz(steps), mfma_probe=ghz(mfma), copy_probe=ghz(copy),
               how='median over the compute units of 100 MHz x d(shader-clock counter) / d(100 MHz counter) between two per-unit stamps on the main stream: 20 steps; 2 MFMA probe launches; 40 copies of 256 MiB')
    del a, b
    return out


def box_probe(dev):
    """What THIS box's GPU sustains on two fixed ~50 ms micro-kernels (csrc/box.hip through rd_box_probe): a 1 GiB streaming copy and a
    dependent-free v_mfma_f32_32x32x16_bf16 loop on every CU.  Boxes of the pool differ by a few per cent (round 5: +-3 %, more than
    a round's gain); `value_normalised` = value / sqrt(copy ratio x mfma ratio) against BOX_REF reads a line through that spread."""
    from ramdsir import _lib as L
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev).fill_(1)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps

    def copy():
        assert lib.rd_box_probe(0, a.data_ptr(), b.data_ptr(), n, st) == 0
    iters = 40000

    def mfma():
        assert lib.rd_box_probe(1, sink.data_ptr(), None, iters, st) == 0
    t_copy = min(timed(copy, 25) for _ in range(3))                  # 25 x 2 GiB of traffic ~ 10 ms each
    t_mfma = min(timed(mfma, 4) for _ in range(3))
    out = dict(copy_gbs=round(2 * n / t_copy / 1e9, 1), mfma_tflops=round(cus * 8 * iters * 4 * 32768 / t_mfma / 1e12, 1), cus=cus,
               device=torch.cuda.get_device_properties(dev).name, ref=dict(BOX_REF),
               how='rd_box_probe: best of 3 x (25 copies of 1 GiB; 4 launches of %d x 4 MFMAs per wave, 8 waves per CU)' % iters)
    out['speed_vs_ref'] = round(((out['copy_gbs'] / BOX_REF['copy_gbs']) * (out['mfma_tflops'] / BOX_REF['mfma_tflops'])) ** 0.5, 4)
    del a, b
    torch.cuda.empty_cache()
    return out


def launch_ranks(gpus):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks ourselves -- `python -m torch.distributed.run
    --nproc-per-node N bench.py <same arguments>` as a CHILD process (never an exec, and before this process has touched the GPU), relay
    what it prints (rank 0's JSON line) and return its exit code.  The reference's counterpart is nn.DataParallel over the visible
    devices (/root/reference/code/train.py:205-208); here it is one process per GPU over RCCL."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != '--launch-ranks']
    env = dict(os.environ, RD_BENCH_SELF_LAUNCHED='1')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    r = subprocess.run(cmd, env=env)
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default='C2', choices=sorted(CONFIGS), help='which of BASELINE.json\'s workloads (C2 = the metric\'s configuration; C3 Prostate 5 x 2 at 384; C5 512 with 4 domains; F256 the reference\'s native Fundus shape)')
    ap.add_argument('--size', type=int, default=0, help='override the side of the configuration (0: its own)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--graph', action='store_true', help='replay one captured hipGraph per step instead of the 3-stream eager launch (slower on ROCm 7: DESIGN.md section 3)')
    ap.add_argument('--no-graph', action='store_true', help='(default; kept for older command lines)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fp32-leg', action='store_true')
    ap.add_argument('--no-pipeline', action='store_true', help='classical step: RAM at the head of every step instead of in the previous step\'s tail')
    ap.add_argument('--no-ablation', action='store_true', help='skip the in-run ablation that picks the headline roofline family')
    ap.add_argument('--no-saturation', action='store_true', help='skip the 4-process run on this GPU that reports gpu_saturated_images_per_s')
    ap.add_argument('--launch-ranks', action='store_true', help='start the ranks as a child torchrun even for --gpus 1 (what --gpus N > 1 does by itself when no launcher set WORLD_SIZE)')
    ap.add_argument('--no-box', action='store_true', help='skip the two micro-kernels that report this box\'s copy rate / MFMA rate (`box`, `value_normalised`)')
    ap.add_argument('--no-live-pmc', action='store_true', help='take the HBM traffic figures from the committed profiles/dominant_kernel_pmc.json instead of measuring them in a rocprofv3 child run')
    args = ap.parse_args()

    if (args.gpus > 1 or args.launch_ranks) and 'WORLD_SIZE' not in os.environ:
        # no launcher around us: start the ranks as a child torchrun (before anything here initialises the GPU) and pass its verdict on.
        # On a box with fewer than N GPUs the child's ranks fail inside RCCL / HIP with the runtime's own message and a non-zero code.
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d runs one rank per GPU but WORLD_SIZE is %d: launch it with `python -m torch.distributed.run '
                         '--nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 bench.py --gpus %d ...`' % (args.gpus, world, args.gpus, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group('nccl', device_id=dev)
    from ramdsir import step as S, ddp as D

    cfg = CONFIGS[args.config]
    bs, Sz, dataset, K = list(cfg['bs']), args.size or cfg['size'], cfg['dataset'], cfg['num_classes']
    B = sum(bs)
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    bank, mods = S.make_bank(dev, 3, 16, K, len(bs))
    init_weights(bank)
    params0 = bank.params.clone()
    # a captured graph replays ONE chain: nothing runs beside the weight gradients, so they keep the whole GPU (tuning.py)
    ts = S.TrainStep(bank, mods, dtype, bs, Sz, Sz, dataset=dataset, consistency=cfg['consistency'], lambda_rec=0.1, lr=cfg['lr'],
                     total_iters=21200, num_classes=K, ram='u8' if dataset == 'fundus' else True,
                     options=dict(side_cus=0, rec_cus=0) if args.graph else None)
    ts.wpack.refresh()
    src, trg, lam, mask, host_inputs = synth_inputs(B, Sz, rank, dev, dataset)
    ts.load_raw(src, trg, lam)
    ts.load_target(mask)
    torch.cuda.synchronize()
    force_ddp = os.environ.get('RD_FORCE_DDP') == '1'          # exercise the 3-graph + bucket path on one GPU
    if force_ddp and world == 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    runner = None
    if world > 1:
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)      # what RCCL sees == what was asked for
    if world > 1 or force_ddp:
        runner = D.DataParallelStep(ts)
        if args.graph:
            runner.capture()
        base_step = runner.step
    else:
        if args.graph:
            ts.capture()
        base_step = ts.step
    if args.graph or args.no_pipeline:
        step = base_step
    else:
        # the pipelined step of train.py (TrainStep.load_raw_next): every step mixes the NEXT batch (RAM) on the restoration lane while
        # its encoder backward runs, and starts on an input that is already mixed.  The synthetic batch is resident in both input slots
        # (inputs in HBM before the timed region, as the contract says); each timed step still runs exactly one RAM mix and one
        # complete training step -- the first one's mix ran during the last warm-up step, the last one mixes for a step
        # that is not timed.
        for dst, val in zip(ts.raw_slots[1], (src, trg, lam)):
            dst.copy_(val)

        def step():
            ts.reuse_next()
            base_step()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    losses = ts.loss_dict()
    assert np.isfinite(losses['loss']), losses
    # data parallel: each gradient bucket's all-reduce on its own (nothing else on the GPU), so that a multi-GPU run says
    # what the exchange costs next to what the step hides of it
    exchange = bucket_allreduce_us(runner, world) if runner is not None else None
    # every rank's lane / queue layout (streams measured to run beside each other, compute-unit budgets), gathered on rank 0
    layouts = [ts.lane_layout()]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, layouts[0])
        layouts = gathered

    if rank == 0:
        out = {
            'metric': 'images/sec (seg+rec+RAM step)', 'value': round(world * B * args.steps / elapsed, 2), 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': cfg['workload'] % (Sz, Sz), 'name': args.config, 'batch_split': bs, 'dataset': dataset, 'consistency': cfg['consistency'], 'lr': cfg['lr'],
                       'global_batch': world * B, 'parallelism': 'dp%d' % world,
                       'process_group': ('%s world %d' % (dist.get_backend(), dist.get_world_size())) if dist.is_initialized() else 'none (single process)', 'hipgraph': bool(args.graph), 'streams': 1 if (args.graph or not ts.fork) else 1 + len(ts.lanes()),
                       'lanes_verified': bool(ts.lanes_verified), 'gradient_exchange': exchange,
                       'lane_layout': layouts, 'ddp_comm': runner.comm_choice if runner is not None else None,
                       'launch_threads': runner.launch_choice if runner is not None else {'launch_threads': bool(ts.launch_threads), 'how': 'single process'},
                       'ram_pipelined': not (args.graph or args.no_pipeline), 'launch': 'rd_run_list (one native call per step)' if not args.graph else 'hipGraph replay',
                       'final_loss': round(losses['loss'], 4)},
        }
        if not args.no_box and not _under_profiler():
            try:
                out['box'] = box_probe(dev)
                out['value_normalised'] = round(out['value'] / out['box']['speed_vs_ref'], 2)
                if world == 1:
                    out['box']['shader_ghz'] = shader_clock(dev, step)
            except Exception as e:                                    # diagnostic: never lose the line over it
                out['box'] = {'error': repr(e)[:200]}
        # HBM traffic measured by this run (a rocprofv3 child of the same command, after the timed region); single process only: under
        # torchrun the committed summary is used
        if world == 1 and runner is None and not args.graph and not args.no_live_pmc:
            collect_live_pmc(B, args.dtype, Sz, args.config)
            if args.config == 'C2':
                collect_live_sq(args.dtype, Sz)
        out['roofline_step'] = whole_step_roofline(B, Sz, out['ms_per_step'], args.dtype)
        if args.dtype == 'bf16':
            roofs = {}
            for fam in FAMILIES:
                roofs[fam] = kernel_roofline(ts, fam, eager=not args.graph)
                if not args.graph:
                    # the same launches on ONE stream (nothing beside them): what the kernels do when they have the GPU to
                    # themselves, next to the headline figures above, which are measured under the step's three-stream contention
                    a = kernel_roofline(ts, fam, eager=False)
                    roofs[fam]['alone'] = {k: a[k] for k in ('achieved', 'unit', 'frac', 'avg_launch_us')}
                mb = mfma_busy(fam)
                if mb is not None:
                    roofs[fam].update(mb)
            # which family is the headline: the three with the most kernel time inside the step, each left out of the step in turn
            by_time = sorted(roofs, key=lambda f: -roofs[f]['step_kernel_us'])
            out['dominant_by_kernel_time'] = {'family': by_time[0], 'step_kernel_us': {f: roofs[f]['step_kernel_us'] for f in by_time}}
            dominant = by_time[0]
            if not args.graph and runner is None and not args.no_ablation:
                cost, base, base2 = step_cost_by_ablation(ts, by_time[:3])
                # the family whose absence shortens the step most -- but costs that tie within 10 % are the same within the measurement
                # (rounds 4-5: 0.54 / 0.52 / 0.52 ms, the headline flipped from run to run): among those the one with the most kernel time
                top = max(cost.values())
                tied = [f for f in by_time[:3] if cost[f] >= 0.9 * top]
                dominant = tied[0]
                out['dominant_by_step_cost'] = {'family': dominant, 'step_ms_saved_without': cost, 'complete_step_ms': [base, base2], 'tied_within_10pct': tied,
                                                'method': 'costs within 10 % of the largest count as tied and the tie goes to the family with the most kernel time; step timed (3 alternating rounds of 20 steps, 3 streams; medians) with the family\'s launches left out; complete_step_ms = [min, max] of the complete step between the rounds'}
                for f in cost:
                    roofs[f]['step_cost_ms'] = cost[f]
            for fam, r in roofs.items():
                out['roofline' if fam == dominant else 'roofline_' + fam] = r
        if world == 1 and args.dtype == 'bf16' and not args.no_fp32_leg:
            out['extra'] = {'fp32': fp32_leg(params0, bs, Sz, dev, src, trg, lam, mask, dataset=dataset, num_classes=K, consistency=cfg['consistency'], lr=cfg['lr'])}
        if world == 1 and runner is None and args.config == 'C2' and Sz == 400 and args.dtype == 'bf16' and not args.no_saturation and not _under_profiler() \
                and os.environ.get('RD_BENCH_CHILD') != '1':
            # how much of the gap to the roofs is dependency bubbles and how much is kernels: the same step from 4 processes at once on this GPU
            try:
                sys.path.insert(0, os.path.join(ROOT, 'scripts'))
                import host_contention as HC
                sat = HC.saturated_rate()
                out['gpu_saturated_images_per_s'] = sat['images_per_s'] if sat else None
                out['gpu_saturated'] = sat
            except Exception as e:                                    # the figure is diagnostic: never lose the line over it
                out['gpu_saturated_images_per_s'] = None
                out['gpu_saturated'] = {'error': repr(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(host_inputs, bs, dataset, K, cfg['consistency'], cfg['lr'])
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
