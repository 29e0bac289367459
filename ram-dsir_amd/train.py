#!/usr/bin/env python3
"""Drop-in for the reference's code/train.py: same flags, same data layout, same checkpoint files, same
training semantics (code/train.py:195-361, 363-528, 530-601) -- on the fused HIP training step.

    python train.py --data_root ../dataset --dataset fundus --domain_idxs 1,2,3 --test_domain_idx 0 \\
                    --ram --rec --is_out_domain --consistency --consistency_type kd --save_path outdir/fundus/target0
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...      # data parallel (RCCL)

Differences, all deliberate: (i) RAM runs on the GPU per batch (the DataLoader workers only pick the partner
image and lambda); (ii) the step is a static launch list enqueued ahead of the GPU, so the five loss scalars are read every
--log_every iterations instead of forcing a device sync every iteration (train.py:298-304) -- and written, under the
reference's tags, to a TensorBoard event file in <save_path>/log (utils/tfevents.py: tensorboardX is not a dependency);
(iii) multi-GPU is one process per GPU with an RCCL gradient all-reduce instead of nn.DataParallel; (iv) tensorboard image
grids and the source-tree snapshot (train.py:306-329,534-536) are not reproduced.
"""
import argparse
import os
import os.path as osp
import random
import time
import sys
from itertools import cycle

HERE = osp.dirname(osp.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

import dataset.transform as trans
from dataset.fundus import Fundus_Multi, Fundus
from dataset.prostate import Prostate_Multi
from networks.unet import Encoder, Decoder, Rec_Decoder, count_params
from utils.metrics import postprocessing, dice_coeff_2label, post_and_dice
from utils.tfevents import SummaryWriter

fundus_batch_list = [[3, 6, 7], [2, 7, 7], [2, 4, 10], [2, 4, 10]]              # train.py:35-38
prostate_batch_list = [[2, 2, 2, 2, 2]] * 6                                       # train.py:40-45


class Compose(object):
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, s):
        for t in self.ts:
            s = t(s)
        return s


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='DG Medical Segmentation Train')
    p.add_argument('--data_root', type=str, default='../dataset')
    p.add_argument('--dataset', type=str, default='fundus', choices=['fundus', 'prostate'])
    p.add_argument('--batch_size', type=int, default=8, help='parsed but unused, as in the reference (train.py:35-45)')
    p.add_argument('--test_batch_size', type=int, default=8)
    p.add_argument('--lr', type=float, default=None)
    p.add_argument('--epochs', type=int, default=None)
    p.add_argument('--domain_idxs', type=str, default='0,1,2')
    p.add_argument('--test_domain_idx', type=int, default=3)
    p.add_argument('--in_channels', type=int, default=3)
    p.add_argument('--num_classes', type=int, default=None)
    p.add_argument('--seed', type=int, default=1337)
    p.add_argument('--lambda_rec', type=float, default=0.1)
    p.add_argument('--deterministic', action='store_true')
    p.add_argument('--ram', action='store_true')
    p.add_argument('--rec', action='store_true')
    p.add_argument('--is_out_domain', action='store_true')
    p.add_argument('--consistency', action='store_true')
    p.add_argument('--consistency_type', type=str, default='mse')
    p.add_argument('--save_path', type=str, default=None, required=True)
    p.add_argument('--norm', type=str, default='bn')
    p.add_argument('--activation', type=str, default='relu')
    p.add_argument('--gpu', type=str, default='0')
    # additions
    p.add_argument('--dtype', type=str, default=None, choices=['bf16', 'f32'],
                   help='activation storage type; default: bf16 for the fused step (--norm bn), f32 for the module-level loop of --norm gn / in '
                        '(its modules are process-wide objects that validation shares: fp32 is their parity-grade mode)')
    p.add_argument('--log_every', type=int, default=20)
    p.add_argument('--num_workers', type=int, default=8)
    p.add_argument('--max_iters', type=int, default=None, help='stop early (smoke runs)')
    return p.parse_args(argv)


def worker_cap(requested, world, n_domains, cpus=None):
    """DataLoader workers per domain loader of one rank: min(requested, cores // (domains x ranks)), at least 1 unless 0 was
    asked for (single-process loading)."""
    if requested <= 0:
        return 0
    cpus = cpus or os.cpu_count() or 8
    return max(1, min(requested, cpus // max(1, n_domains * world)))


def seed_worker(worker_id):
    worker_seed = torch.initial_seed() % 2 ** 32
    np.random.seed(worker_seed)
    random.seed(worker_seed)


_VAL = {}


def _val_resources(data_dir, datasetTest, batch_size):
    """The test loader and the post-processing pool live across epochs: the reference rebuilds an 8-worker loader and
    post-processes every 800x800 prediction serially at every epoch (~0.2 s per image: connected components + hole filling) --
    seconds per epoch, invisible next to its training time, most of the wall clock next to this one's."""
    key = (data_dir, datasetTest, batch_size)
    if key not in _VAL:
        import multiprocessing as mp
        testset = Fundus(base_dir=data_dir, split='test', domain_idx=datasetTest,
                         transform=Compose([trans.Resize((256, 256)), trans.Normalize()]))
        nw = min(8, max(1, len(testset) // 2))
        loader = DataLoader(testset, batch_size=batch_size, num_workers=nw, shuffle=False, drop_last=False, pin_memory=True,
                            persistent_workers=True)
        pool = mp.get_context('fork').Pool(min(16, max(2, (os.cpu_count() or 4) // 4)))     # children never touch the GPU
        # a DataLoader forks its workers at the first iter(), not at construction: start them NOW (main() calls this before the
        # first GPU call), so that they too are forked from an address space that has never initialised HIP.  With
        # persistent_workers the DataLoader keeps this iterator (and its workers) and resets it at every later `for ... in loader`.
        it = iter(loader)
        _VAL[key] = (loader, pool, it)
    return _VAL[key][:2]


def _close_val():
    for loader, pool, it in _VAL.values():
        pool.terminate()
        pool.join()
        del it, loader                                      # persistent workers shut down with the loader's iterator
    _VAL.clear()


def test_fundus(encoder, seg_decoder, epoch, data_dir, datasetTest, output_path, batch_size=8, dataset='fundus'):
    """train.py:91-132 (BN in eval mode here, as in the reference's in-training evaluation): sigmoid, bilinear resize to the
    native mask size, threshold 0.75, largest component + hole filling, Dice with +1 smoothing, one CSV line."""
    encoder.eval()
    seg_decoder.eval()
    loader, pool = _val_resources(data_dir, datasetTest, batch_size)
    cup = disc = 0.0
    n = 0
    pending = []
    with torch.no_grad():
        for data, target, target_orig, ids in loader:
            pred = torch.sigmoid(seg_decoder(encoder(data.cuda(non_blocking=True))))
            pred = torch.nn.functional.interpolate(pred, size=(target_orig.size(2), target_orig.size(3)), mode='bilinear')
            masks = (pred > 0.75).to(torch.uint8).cpu().numpy()                  # the threshold of postprocessing(), on the GPU
            tg = target_orig.to(torch.uint8).numpy()
            pending.append(pool.map_async(post_and_dice, [(masks[i], tg[i]) for i in range(masks.shape[0])]))
    for r in pending:
        for c, d in r.get():
            cup, disc, n = cup + c, disc + d, n + 1
    cup, disc = cup / max(n, 1), disc / max(n, 1)
    print('val_cup_dice : {}, val_disc_dice : {}'.format(cup, disc))
    with open(osp.join(output_path, str(datasetTest) + '_val_log.csv'), 'a') as f:
        f.write(','.join(map(str, [['batch-size: '] + [batch_size] + [epoch] + ['cup dice coefficence: '] + [cup] +
                                   ['disc dice coefficence: '] + [disc]])) + '\n')
    return (cup + disc) * 100.0 / 2


def test_prostate(encoder, seg_decoder, epoch, data_dir, datasetTest, output_path, batch_size=8, dataset='prostate'):
    """train.py:134-192: Dice over the NIfTI volumes of the held-out site (BN in eval mode), one line in
    <target>_val_log.csv.  `data_dir` is the dataset directory (<data_root>/prostate)."""
    from utils.prostate_eval import DOMAIN_LIST, evaluate_domain
    encoder.eval()
    seg_decoder.eval()
    with torch.no_grad():
        val_dice, _, _ = evaluate_domain(lambda v: seg_decoder(encoder(v.cuda())), data_dir, DOMAIN_LIST[datasetTest], batch_size)
    print('val_dice : {}'.format(val_dice))
    with open(osp.join(output_path, str(datasetTest) + '_val_log.csv'), 'a') as f:
        f.write(','.join(map(str, [['batch-size: '] + [batch_size] + [epoch] + ['dice coefficence: '] + [val_dice]])) + '\n')
    return val_dice * 100.0


def save_checkpoint(path, encoder, seg_decoder, rec_decoder):
    """train.py:342-360: unwrapped state_dicts under the reference's three keys."""
    torch.save({'encoder_state_dict': encoder.state_dict(), 'seg_decoder_state_dict': seg_decoder.state_dict(),
                'rec_decoder_state_dict': rec_decoder.state_dict()}, path)


def main(args):
    if not (args.ram and args.rec):
        # the reference only runs end-to-end with --ram --rec (train.py:591 / :315 raise otherwise)
        raise ValueError('only the --ram --rec flag combination exists in the reference; got ram=%s rec=%s' % (args.ram, args.rec))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    data_root = os.path.join(args.data_root, args.dataset)
    # The validation pool (fork) and the persistent test loader -- with its worker processes started (_val_resources) -- are created
    # HERE, before the first GPU call of this process: their children are forked from an address space that has never initialised
    # HIP (rank 0 validates, below)
    has_val = args.dataset == 'fundus' and os.path.exists(os.path.join(data_root, 'Domain%d_test.list' % (args.test_domain_idx + 1)))
    if rank == 0 and has_val:
        _val_resources(data_root, args.test_domain_idx, args.test_batch_size)
    # DataLoader workers per domain loader: the reference's 8 (train.py:558) on one GPU; under torchrun every rank has its own
    # three loaders, so the host's cores are divided among ranks x domains (8 ranks x 3 x 8 = 192 decoder processes otherwise)
    n_dom = len(args.domain_idxs.split(','))
    args.num_workers = worker_cap(args.num_workers, world, n_dom)
    torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    # host-side rendezvous (gloo) for the points where ranks wait for rank 0 on the CPU (validation): an RCCL barrier is a GPU
    # kernel that spins until every peer has arrived
    host_group = dist.new_group(backend='gloo') if world > 1 else None
    os.makedirs(args.save_path, exist_ok=True)

    zoo = {'fundus': Fundus_Multi, 'prostate': Prostate_Multi}
    transform = {'fundus': Compose([trans.Resize((256, 256)), trans.RandomScaleCrop((256, 256))]), 'prostate': None}
    bsl = fundus_batch_list[args.test_domain_idx] if args.dataset == 'fundus' else prostate_batch_list[args.test_domain_idx]
    domain_idx_list = [int(i) for i in args.domain_idxs.split(',')]
    loaders, max_len, max_id = [], -1, 0
    raw, samplers = [], []
    for idx, i in enumerate(domain_idx_list):
        ds = zoo[args.dataset](base_dir=data_root, split='train', domain_idx_list=[i], transform=transform[args.dataset],
                               is_out_domain=args.is_out_domain, test_domain_idx=args.test_domain_idx)
        # Data parallel (one process per GPU): every domain's list is SHARDED over the ranks (DistributedSampler, reshuffled
        # per epoch with seed + epoch), each rank draws the reference's per-domain batch sizes from its shard, so one step
        # consumes world x the reference's batch and an epoch is 1/world as many iterations; RAM partners / lambda / crops
        # are drawn rank-locally (workers are seeded from seed + rank).  Single process: exactly the reference's loaders.
        sampler = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=args.seed, drop_last=True) if world > 1 else None
        # the reference's loader settings (train.py:558-559) + persistent workers with a deeper queue: the reference respawns its
        # 3 x 8 worker processes at every epoch, which costs seconds per 53-iteration epoch -- invisible next to its step time,
        # 20x the step time here (profiles/README.md, end-to-end throughput)
        extra = dict(persistent_workers=True, prefetch_factor=4) if args.num_workers > 0 else {}
        dl = DataLoader(ds, batch_size=bsl[idx], num_workers=args.num_workers, shuffle=sampler is None, sampler=sampler, drop_last=True,
                        pin_memory=True, worker_init_fn=seed_worker, **extra)
        raw.append(dl)
        samplers.append(sampler)
        loaders.append(cycle(dl))                       # train.py:560: replays the first pass of the shorter loaders
        if max_len < len(dl):
            max_len, max_id = len(dl), idx
    loaders[max_id] = raw[max_id]

    encoder = Encoder(c=args.in_channels, norm=args.norm, activation=args.activation).cuda()
    seg_decoder = Decoder(num_classes=args.num_classes, norm=args.norm, activation=args.activation).cuda()
    rec_decoder = Rec_Decoder(num_classes=args.in_channels, norm='dsbn', activation=args.activation,
                              num_domains=len(domain_idx_list)).cuda()
    print('\nEncoder Params: %.3fM' % count_params(encoder))
    print('\nSeg Decoder Params: %.3fM' % count_params(seg_decoder))
    print('\nRec Decoder Params: %.3fM' % count_params(rec_decoder))

    from ramdsir.trainer import FusedTrainer, ModuleTrainer
    sample = next(iter(raw[0]))
    H, W = sample[0].shape[1:3]
    total_iters = max_len * args.epochs
    cons = args.consistency_type if args.consistency else None
    assert cons in (None, 'mse', 'kd'), args.consistency_type
    # --norm bn (the reference's default): the fused HIP step.  gn / in: the reference's own loop over the drop-in modules (the fused step
    # keeps the statistics groups of the shared BatchNorms and of the restoration decoder's DSBN in one launch list: bn only)
    Trainer = FusedTrainer if args.norm == 'bn' else ModuleTrainer
    if args.norm != 'bn' and rank == 0:
        print('norm=%s: module-level training loop (torch autograd between the HIP modules)' % args.norm)
    trainer = Trainer(encoder, seg_decoder, rec_decoder, bsl[:len(domain_idx_list)], H, W, dataset=args.dataset,
                           consistency=cons, lambda_rec=args.lambda_rec, lr=args.lr, total_iters=total_iters,
                           dtype=torch.bfloat16 if (args.dtype or ('bf16' if args.norm == 'bn' else 'f32')) == 'bf16' else torch.float32)

    writer = SummaryWriter(os.path.join(args.save_path, 'log')) if rank == 0 else None      # train.py:538
    previous_best, iter_num = 0.0, 0
    t_mark, it_mark, imgs_per_iter = None, 0, world * sum(bsl[:len(domain_idx_list)])
    for epoch in range(args.epochs):
        it_epoch = iter_num
        if rank == 0:
            print('\n==> Epoch %i, learning rate = %.6f' % (epoch, args.lr if iter_num == 0 else trainer.lr()))
        for m in (encoder, seg_decoder, rec_decoder):
            m.train()
        t_epoch = time.time()
        for sp in samplers:
            if sp is not None:
                sp.set_epoch(epoch)
        def on_device(batches):
            return tuple(torch.cat([b[k] for b in batches], 0).cuda(non_blocking=True) for k in range(4))       # src, trg, lam, mask

        # one batch of look-ahead: the step of batch i also mixes batch i+1 (RAM) in its tail, beside Adam and the weight repack
        # (FusedTrainer.step next_batch; the last iteration of an epoch -- or of --max_iters -- has no successor and runs the
        # classical step)
        stream_it = iter(zip(*loaders))
        cur = next(stream_it, None)
        cur = on_device(cur) if cur is not None else None
        i = -1
        while cur is not None:
            i += 1
            nxt = next(stream_it, None)
            last = nxt is None or bool(args.max_iters and iter_num + 1 >= args.max_iters)
            nxt = None if last else on_device(nxt)
            trainer.step(*cur, next_batch=nxt)
            cur = nxt
            if iter_num % args.log_every == 0:
                l = trainer.losses()                    # collective when world > 1: the mean over the ranks
            if rank == 0 and iter_num % args.log_every == 0:
                lr_now = trainer.lr()
                print('iter %d lr %.6f ' % (iter_num, lr_now) + ' '.join('%s %.4f' % (k, v) for k, v in l.items() if k != 'rec')
                      + ' loss_rec %.4f' % (sum(l['rec']) / 4))               # train.py:304 logs avg/4
                # the reference's scalars (train.py:298-304 / 467-473), at the iterations whose losses are read back
                writer.add_scalars_at(iter_num, [('lr', lr_now)] + [('loss/' + k, v) for k, v in l.items() if k not in ('rec', 'loss')]
                                      + [('loss/loss_rec', sum(l['rec']) / 4)])
            iter_num += 1
            if iter_num == 5:                           # end-to-end throughput (files -> DataLoader -> H2D -> step), start-up excluded
                torch.cuda.synchronize()
                t_mark, it_mark = time.time(), iter_num
            if args.max_iters and iter_num >= args.max_iters:
                break
        if t_mark is not None and iter_num > it_mark:
            torch.cuda.synchronize()
            if rank == 0:
                print('train throughput: %.1f images/s end to end (%d iterations of %d images, %d DataLoader workers per domain)'
                      % ((iter_num - it_mark) * imgs_per_iter / (time.time() - t_mark), iter_num - it_mark, imgs_per_iter, args.num_workers))
        torch.cuda.synchronize()
        t_train = time.time() - t_epoch
        # validation on the held-out domain + keep-best checkpoint rotation (train.py:331-350); skipped when the
        # evaluation data is not on disk (synthetic / smoke runs)
        avg_dice = None
        if rank == 0 and args.dataset == 'fundus' and os.path.exists(os.path.join(data_root, 'Domain%d_test.list' % (args.test_domain_idx + 1))):
            print('Test on target domain {}'.format(args.test_domain_idx))
            avg_dice = test_fundus(encoder, seg_decoder, epoch, data_root, args.test_domain_idx, args.save_path, args.test_batch_size)
        elif rank == 0 and args.dataset == 'prostate':
            from utils.prostate_eval import DOMAIN_LIST
            if os.path.isdir(os.path.join(data_root, DOMAIN_LIST[args.test_domain_idx])):
                print('Test on target domain {}'.format(args.test_domain_idx))
                avg_dice = test_prostate(encoder, seg_decoder, epoch, data_root, args.test_domain_idx, args.save_path, args.test_batch_size)
        if world > 1:
            # every rank has finished its epoch (synchronize above) and none has entered the next step's all-reduce: the
            # other ranks wait HERE, on the host (gloo), while rank 0 validates with replica 0's BatchNorm statistics (what
            # nn.DataParallel evaluates, train.py:343) -- not inside an RCCL gradient exchange with a missing peer, which
            # spins their GPUs and, past the watchdog timeout, aborts the job
            dist.barrier(group=host_group)
        if avg_dice is not None and avg_dice >= previous_best:
            if previous_best != 0:
                old = os.path.join(args.save_path, 'model_%.2f.pth' % previous_best)
                if os.path.exists(old):
                    os.remove(old)
            save_checkpoint(os.path.join(args.save_path, 'model_%.2f.pth' % avg_dice), encoder, seg_decoder, rec_decoder)
            previous_best = avg_dice
        if rank == 0:
            t_all, n_img = time.time() - t_epoch, (iter_num - it_epoch) * imgs_per_iter
            print('epoch %d: training %.2f s (%.0f images/s), validation + checkpoint %.2f s (%.0f images/s over the whole epoch)'
                  % (epoch, t_train, n_img / max(t_train, 1e-9), t_all - t_train, n_img / max(t_all, 1e-9)))
        if args.max_iters and iter_num >= args.max_iters:
            break
    if rank == 0:
        save_checkpoint(os.path.join(args.save_path, 'final_model.pth'), encoder, seg_decoder, rec_decoder)
        print('\nSave Final Model to {}'.format(args.save_path))
    if writer is not None:
        writer.close()
    _close_val()
    if world > 1:
        dist.barrier(group=host_group)


if __name__ == '__main__':
    args = parse_args()
    if 'LOCAL_RANK' not in os.environ:
        os.environ['CUDA_VISIBLE_DEVICES'] = args.gpu                  # train.py:606
    if args.deterministic:
        r = int(os.environ.get('RANK', '0'))               # rank-local RAM partners / crops; weights are broadcast from rank 0
        random.seed(args.seed + r)
        np.random.seed(args.seed + r)
        torch.manual_seed(args.seed + r)
    if args.epochs is None:
        args.epochs = {'fundus': 400, 'prostate': 200}[args.dataset]
    if args.lr is None:
        args.lr = {'fundus': 2e-3, 'prostate': 1e-3}[args.dataset]
    if args.num_classes is None:
        args.num_classes = {'fundus': 2, 'prostate': 2}[args.dataset]
    print(args)
    main(args)
