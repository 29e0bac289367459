"""-m gpu: bench.py's data-parallel path on one GPU (RD_FORCE_DDP=1: a one-rank RCCL process group), so that the four-segment launch
with the three gradient-bucket exchanges (ramdsir/ddp.py; reference: nn.DataParallel's gradient reduction, code/train.py:205-208) is
run and TIMED on every driver round even when no multi-GPU node is available, and the bench line's multi-GPU fields are exercised."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_on_the_forced_data_parallel_path():
    env = dict(os.environ, RD_FORCE_DDP='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29541')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '6', '--warmup', '3', '--no-cpu-baseline', '--no-fp32-leg',
                        '--no-ablation'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    cfg = d['config']
    assert d['n_gpus'] == 1 and d['steps'] == 6 and d['value'] > 0 and d['unit'] == 'images/s'
    assert cfg['process_group'] == 'nccl world 1' and cfg['parallelism'] == 'dp1'
    ex = cfg['gradient_exchange']                                   # decoders, encoder levels 3-5, encoder levels 1-2, in launch order
    assert [e['bucket'] for e in ex] == [2, 1, 0] and sum(e['bytes'] for e in ex) == 4 * 3800021
    assert all(e['allreduce_us'] > 0 for e in ex)
    assert cfg['ddp_comm']['own_comm_stream'] is False              # one rank: nothing to measure, the weight-gradient lane
    lt = cfg['launch_threads']                                      # one host thread or one per lane: measured at start-up (ddp.py)
    assert lt['how'] == 'measured' and lt['ms_per_step_one_thread'] > 0 and lt['ms_per_step_lane_threads'] > 0
    lay = cfg['lane_layout']
    assert len(lay) == 1 and set(lay[0]) >= {'main', 'side0', 'rec', 'budgets'} and lay[0]['budgets']['side_cus'] == 96
    assert cfg['ram_pipelined'] is True and abs(cfg['final_loss']) < 10
    # the exchange path costs little on one rank: within 15 % of the plain step of the same run's roofline block
    assert d['roofline']['bound'] in ('hbm', 'mfma') and d['roofline_step']['algorithmic_bytes_per_step'] > 8e9


@pytest.mark.parametrize('name,dataset,split,side', [('C3', 'prostate', [2, 2, 2, 2, 2], 384), ('C5', 'fundus', [2, 2, 2, 2], 512),
                                                     ('F256', 'fundus', [3, 6, 7], 256)])
def test_bench_lines_of_the_other_baseline_configs(name, dataset, split, side):
    """bench.py --config: the workloads BASELINE.json lists beside the metric's own (C3 Prostate 5 x 2 at 384 with softmax / CE / dice_multi,
    train.py:39-45,363-465; C5 512 x 512 with 4 domains; F256 the reference's native Fundus shape, train.py:35,541) give a line with the
    same fields: throughput of the whole step at that shape, its roofline position, the workload named in config."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', name, '--steps', '4', '--warmup', '2', '--no-cpu-baseline',
                        '--no-fp32-leg', '--no-ablation', '--no-live-pmc', '--no-saturation'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith('{')][-1])
    cfg = d['config']
    assert cfg['name'] == name and cfg['dataset'] == dataset and cfg['batch_split'] == split and ('%dx%d' % (side, side)) in cfg['workload']
    assert cfg['global_batch'] == sum(split) and d['value'] > 0 and d['dtype'] == 'bf16' and abs(cfg['final_loss']) < 20
    rs = d['roofline_step']
    assert abs(rs['bytes_per_image'] / (1027560000 * (side / 400.0) ** 2) - 1) < 0.01         # SURVEY.md 8d scaling with the side
    assert d['roofline']['bound'] in ('hbm', 'mfma') and 0 < d['roofline']['frac'] < 1


def test_bench_launches_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus N` with WORLD_SIZE unset starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child
    (bench.launch_ranks: never an exec, before the parent touches the GPU) and relays rank 0's line and the exit code.  --launch-ranks
    forces the same path for N = 1 (this box has one GPU); with --gpus 2 here the ranks must fail inside the runtime / RCCL -- not with
    bench.py's own "WORLD_SIZE is 1" refusal.  Reference parallelism: nn.DataParallel, code/train.py:205-208."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'RD_FORCE_DDP')}
    base = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '2', '--no-cpu-baseline', '--no-fp32-leg', '--no-ablation',
            '--no-live-pmc', '--no-saturation']
    r = subprocess.run(base + ['--gpus', '1', '--launch-ranks'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    d = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 1 and d['value'] > 0 and d['config']['parallelism'] == 'dp1'
    assert d['box']['copy_gbs'] > 1000 and d['box']['mfma_tflops'] > 500 and d['value_normalised'] > 0
    assert all(1.0 < d['box']['shader_ghz'][k] < 3.0 for k in ('step', 'mfma_probe', 'copy_probe')), d['box']['shader_ghz']
    import torch
    if torch.cuda.device_count() < 2:
        r2 = subprocess.run(base + ['--gpus', '2', '--no-box'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        err = r2.stderr.decode()
        assert r2.returncode != 0
        assert 'runs one rank per GPU but WORLD_SIZE is' not in err, err[-2000:]
