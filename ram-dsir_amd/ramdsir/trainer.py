"""Glue between the drop-in nn.Modules and the fused HIP training step: puts the three modules into ONE
parameter arena (encoder first: Adam's lr/2 group, code/train.py:573), builds the TrainStep for the batch
geometry and (multi-GPU) wraps it with the bucketed RCCL gradient exchange.  The step is launched eagerly over three
HIP streams (step.TrainStep.lanes; the host enqueues a step in ~1.4 ms against ~7 ms of GPU time); use_graph=True
replays one captured hipGraph per step instead (no host work per step, but slower on ROCm 7: DESIGN.md section 3)."""
import torch
import torch.distributed as dist

from . import engine as E
from . import step as S
from . import ddp as D


class FusedTrainer:
    def __init__(self, encoder, seg_decoder, rec_decoder, batch_sizes, H, W, dataset='fundus', consistency='kd', lambda_rec=0.1,
                 lr=2e-3, total_iters=1, dtype=torch.bfloat16, use_graph=False, device=None):
        dev = torch.device('cuda', torch.cuda.current_device()) if device is None else device
        self.modules = dict(enc=encoder, dec=seg_decoder, rec=rec_decoder)
        mods = [('enc', encoder._specs), ('dec', seg_decoder._specs), ('rec', rec_decoder._specs)]
        self.bank = E.ParamBank(mods, dev)
        for name, m in self.modules.items():
            m.bind(self.bank, name, None)
        slope = encoder._slope
        self.ts = S.TrainStep(self.bank, mods, dtype, batch_sizes, H, W, dataset=dataset, consistency=consistency,
                              lambda_rec=lambda_rec, lr=lr, total_iters=total_iters, in_channels=encoder._c, n=encoder._n,
                              num_classes=seg_decoder._k, slope=slope, ram='u8' if dataset == 'fundus' else True,
                              # a captured graph replays one chain: no lane to leave compute units to (tuning.py)
                              options=dict(side_cus=0, rec_cus=0) if use_graph else None)
        self.ts.wpack.refresh()
        self._primed, self._graphs, self._announced = False, bool(use_graph), None
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        if self.world > 1:
            # identical initial weights on every rank (DataParallel broadcasts replica 0's, train.py:205-208)
            dist.broadcast(self.bank.params, src=0)
            for b in self.bank.buffers.values():
                dist.broadcast(b, src=0)
            self.ts.wpack.refresh()
            self.runner = D.DataParallelStep(self.ts)
            if use_graph:
                self.runner.capture()
            self._step = self.runner.step
        else:
            if use_graph:
                self.ts.capture()
            self._step = self.ts.step

    def step(self, src_nhwc, trg_nhwc, lam, target, next_batch=None):
        """src/trg: NHWC images as the dataset holds them before RAM -- fundus: uint8 pixels 0..255 (the step is built with
        uint8 RAM buffers; float images are refused by TrainStep.load_raw), prostate: float32 in [-1,1];
        lam: [B]; target: fundus (B,2,H,W) float multilabel mask / prostate (B,H,W) int64.
        next_batch = (src, trg, lam, target) of the FOLLOWING step, if the caller has it: its images go into the other input slot now
        and are mixed (RAM) in this step's tail, beside Adam and the weight repack, instead of opening the next step alone
        (step.TrainStep.load_raw_next); the next call's first four arguments are then ignored -- that batch is already on the
        device.  Not with captured graphs (they replay the classical step)."""
        if not self._primed:
            self.ts.load_raw(src_nhwc, trg_nhwc, lam)
            self.ts.load_target(target)
        elif not all(a is b for a, b in zip((src_nhwc, trg_nhwc, lam, target), self._announced)):
            # the previous call uploaded `next_batch` and this step trains on THAT: a caller that now passes other tensors would
            # silently train on the wrong data
            raise ValueError('FusedTrainer.step: the previous call announced next_batch; this call must pass the same tensors '
                             '(src, trg, lam, target) as its current batch')
        pipelined = next_batch is not None and not self._graphs
        if pipelined:
            self.ts.load_raw_next(*next_batch[:3])
        self._step()
        if pipelined:
            self.ts.load_target(next_batch[3])            # after the step has been enqueued: its loss still reads the current mask
        self._primed = pipelined
        self._announced = tuple(next_batch[:4]) if pipelined else None

    def losses(self):
        """The five loss terms + per-domain rec losses (names of the tensorboard scalars, train.py:298-304).  Data parallel:
        the MEAN over the ranks (every rank must call this; one small all-reduce at logging time)."""
        if self.world > 1:
            buf = torch.cat([self.ts.losses, self.ts.rec_mse])
            dist.all_reduce(buf)
            buf /= self.world
            keep = self.ts.losses.clone(), self.ts.rec_mse.clone()
            self.ts.losses.copy_(buf[:self.ts.losses.numel()])
            self.ts.rec_mse.copy_(buf[self.ts.losses.numel():])
            out = self.ts.loss_dict()
            self.ts.losses.copy_(keep[0])
            self.ts.rec_mse.copy_(keep[1])
            return out
        return self.ts.loss_dict()

    def lr(self):
        return float(self.ts.hyper[0])


class ModuleTrainer:
    """The reference's own training step (code/train.py:246-296 fundus, 412-465 prostate) over the drop-in modules of networks/unet.py
    -- torch autograd between the modules, torch's losses and torch.optim.Adam, the poly schedule of train.py:289-293 -- with the
    FusedTrainer interface.  train.py takes this path for --norm gn / in: the fused step keeps the statistics groups of the shared
    BatchNorms and of the restoration decoder's DSBN in one launch list and implements `bn` only, the modules implement all of
    normalization().  The convolutions / normalisations are the HIP launch lists of the modules, RAM is rd_ram_mix; single process."""

    def __init__(self, encoder, seg_decoder, rec_decoder, batch_sizes, H, W, dataset='fundus', consistency='kd', lambda_rec=0.1,
                 lr=2e-3, total_iters=1, dtype=torch.bfloat16, **_):
        if dist.is_initialized() and dist.get_world_size() > 1:
            raise NotImplementedError('norm gn / in: the module-level training loop runs in one process (no gradient exchange)')
        from torch.optim import Adam
        self.enc, self.dec, self.rec = encoder, seg_decoder, rec_decoder
        self.bs, self.dataset, self.cons, self.lambda_rec = list(batch_sizes), dataset, consistency, lambda_rec
        self.base_lr, self.total_iters, self.iter_num = lr, total_iters, 0
        self.num_classes = seg_decoder._k
        self.opt = Adam([{'params': encoder.parameters(), 'lr': lr / 2}, {'params': seg_decoder.parameters(), 'lr': lr},
                         {'params': rec_decoder.parameters(), 'lr': lr}], lr=lr, betas=(0.9, 0.999))                 # train.py:573-576
        self._last = None
        # the modules' activation storage follows `dtype` (the default of train.py --dtype is bf16; the RAMDSIR_DTYPE environment default
        # of the module path is fp32: without this the flag was silently ignored here)
        from . import modules as M
        M.set_storage_dtype(dtype)
        self.dtype = dtype
        # ONE RAM plan (twiddles, workspace, the two output tensors), bound per step: the geometry is fixed
        from .ram import RamMixer
        B = sum(self.bs)
        dev = next(encoder.parameters()).device
        self._ram = RamMixer(B, H, W, torch.float32, dev, dataset)
        self._ram_out = (torch.empty(B, H, W, 3, dtype=torch.float32, device=dev), torch.empty(B, H, W, 3, dtype=torch.float32, device=dev))

    def _mix(self, src_nhwc, trg_nhwc, lam):
        """(img, img_freq) NCHW fp32 in [-1, 1] (ram.source_to_target_freq_batch with the plan built once)."""
        if not (src_nhwc.dtype == torch.uint8 and trg_nhwc.dtype == torch.uint8):
            src_nhwc, trg_nhwc = src_nhwc.float(), trg_nhwc.float()
        oi, of = self._ram_out
        self._ram.bind(src_nhwc.contiguous(), trg_nhwc.contiguous(), lam.float().contiguous(), oi, of)
        self._ram.run()
        return oi.permute(0, 3, 1, 2).contiguous(), of.permute(0, 3, 1, 2).contiguous()

    def step(self, src_nhwc, trg_nhwc, lam, target, next_batch=None):
        import torch.nn.functional as F
        from utils.losses import dice_loss, dice_loss_multi
        img, img_freq = self._mix(src_nhwc, trg_nhwc, lam)
        fundus = self.dataset == 'fundus'

        def seg(feats):
            logits = self.dec(feats)
            if fundus:
                soft = torch.sigmoid(logits)
                return soft, F.binary_cross_entropy(soft, target), dice_loss(soft, target)
            soft = torch.softmax(logits, dim=1)
            return soft, F.cross_entropy(logits, target), dice_loss_multi(soft, target, num_classes=self.num_classes, ignore_index=0)

        soft1, seg1, dice1 = seg(self.enc(img))
        feats2 = self.enc(img_freq)
        soft2, seg2, dice2 = seg(feats2)
        zero = torch.zeros((), device=img.device)
        if self.cons == 'kd':                                                                   # train.py:85-88 KD(pred_soft_2, pred_soft_1)
            consistency = F.kl_div(soft2.log(), soft1, reduction='mean') + F.kl_div(soft1.log(), soft2, reduction='mean')
        elif self.cons == 'mse':
            consistency = F.mse_loss(soft2, soft1)
        else:
            consistency = zero
        loss, left, rec_l = 0, 0, []
        for d, b in enumerate(self.bs):                                                         # train.py:265-276
            rec_soft = torch.tanh(self.rec(feats2[-1][left:left + b], domain_label=d * torch.ones(b, dtype=torch.long)))
            l = F.mse_loss(rec_soft, img[left:left + b])
            loss = loss + self.lambda_rec * l
            rec_l.append(l)
            left += b
        loss = loss + seg1 + seg2 + dice1 + dice2 + 0.5 * consistency
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        lr = self.base_lr * (1 - self.iter_num / self.total_iters) ** 0.9                       # train.py:289-293
        self.opt.param_groups[0]['lr'], self.opt.param_groups[1]['lr'], self.opt.param_groups[2]['lr'] = lr / 2, lr, lr
        self.iter_num += 1
        self._last = (seg1, dice1, seg2, dice2, consistency, loss, rec_l)

    def losses(self):
        seg1, dice1, seg2, dice2, cons, loss, rec_l = self._last
        s = 'bce' if self.dataset == 'fundus' else 'ce'
        return {'loss_%s_1' % s: float(seg1), 'loss_dice_1': float(dice1), 'loss_%s_2' % s: float(seg2), 'loss_dice_2': float(dice2),
                'loss_consistency': float(cons), 'rec': [float(r) for r in rec_l], 'loss': float(loss)}

    def lr(self):
        return float(self.opt.param_groups[1]['lr'])
