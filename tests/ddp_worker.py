"""One rank of the 2-process data-parallel test (tests/test_gpu_ddp.py): both ranks share cuda:0, the process group is gloo
(on device tensors), so ramdsir.ddp.DataParallelStep -- segments A / B1 / B2 / C, three bucket exchanges on the communication
stream, Adam after the join -- runs with world_size 2 without a second GPU.  Prints one JSON line.  TEST INFRASTRUCTURE."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'ram-dsir_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rank', type=int, required=True)
    ap.add_argument('--world', type=int, default=2)
    ap.add_argument('--port', type=int, required=True)
    ap.add_argument('--graph', type=int, default=0)
    ap.add_argument('--size', type=int, default=64)
    ap.add_argument('--backend', default='gloo', choices=['gloo', 'nccl'])
    args = ap.parse_args()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(args.port))
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    if args.backend == 'nccl':
        # RCCL: one rank per device (world 1 on a one-GPU box, rank r on cuda:r otherwise) -- the asynchronous AVG branch of
        # ramdsir.ddp.Buckets, which gloo never takes
        dev = torch.device('cuda', args.rank)
        torch.cuda.set_device(dev)
        dist.init_process_group('nccl', rank=args.rank, world_size=args.world, device_id=dev)
    else:
        dist.init_process_group('gloo', rank=args.rank, world_size=args.world)
    import fullsize_util as FU
    from ramdsir import step as S, ddp as D
    cfg = dict(dataset='fundus', bs=[1, 2, 1], S=args.size)
    states = FU.oracle_states(3)

    def make(seed):
        src, trg, lam, mask = FU.synth(cfg, seed=seed)
        bank, mods = S.make_bank(dev, 3, 16, 2, 3)
        for m, sd in zip(('enc', 'dec', 'rec'), states):
            S.load_state(bank, m, sd)
        ts = S.TrainStep(bank, mods, torch.float32, cfg['bs'], args.size, args.size, dataset='fundus', consistency='kd', lr=2e-3,
                         total_iters=100, ram=True)
        ts.wpack.refresh()
        T = lambda a: torch.from_numpy(a).to(dev)
        ts.load_raw(T(src), T(trg), T(lam))
        ts.load_target(T(mask))
        return ts, bank

    # (1) this rank's own gradient, from a plain single-process step on ITS batch (seed 1337 + rank)
    ts0, bank0 = make(1337 + args.rank)
    ts0.zero()
    ts0.run_segment(ts0.seg_a + ts0.seg_b)              # forward + backward, no Adam
    torch.cuda.synchronize()
    g_local = bank0.grads.clone()
    l_local = ts0.losses[:5].clone()
    # (2) the data-parallel step on the same batch
    ts, bank = make(1337 + args.rank)
    runner = D.DataParallelStep(ts)
    assert runner.buckets.world == args.world
    if args.graph:
        runner.capture()
    p_before = bank.params.clone()
    runner.step()
    torch.cuda.synchronize()
    g_avg = bank.grads.clone()                          # the arena holds the averaged gradient after the exchange
    # expected: mean over ranks of the local gradients
    gl = [torch.empty_like(g_local) for _ in range(args.world)]
    dist.all_gather(gl, g_local)
    g_mean = sum(gl) / args.world
    rel = float((g_avg.double() - g_mean.double()).norm() / g_mean.double().norm())
    rel_self = float((g_avg.double() - g_local.double()).norm() / g_local.double().norm())   # must NOT be ~0: ranks differ
    ga = [torch.empty_like(g_avg) for _ in range(args.world)]
    dist.all_gather(ga, g_avg)
    same_grad = all(torch.equal(ga[0], t) for t in ga)
    runner.step()                                       # second step on the same batches
    torch.cuda.synchronize()
    pl = [torch.empty_like(bank.params) for _ in range(args.world)]
    dist.all_gather(pl, bank.params)
    same_params = all(torch.equal(pl[0], t) for t in pl)
    moved = float((bank.params - p_before).abs().max())
    # losses: rank-local values and their all-reduced mean (what train.py logs)
    lm = ts.losses[:5].clone()
    dist.all_reduce(lm)
    lm /= args.world
    out = dict(rank=args.rank, rel_avg_vs_mean=rel, rel_avg_vs_local=rel_self, same_grad=bool(same_grad), same_params=bool(same_params),
               moved=moved, iters=int(ts.iter), loss_local=[float(v) for v in l_local], loss_mean=[float(v) for v in lm],
               bn_tracked=int(bank.b('enc', 'convd1.bn1.num_batches_tracked')), graph=args.graph, backend=dist.get_backend(),
               avg_op=str(runner.buckets._avg))
    print('DDPRESULT ' + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
