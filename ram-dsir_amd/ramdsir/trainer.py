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

    def step(self, src_nhwc, trg_nhwc, lam, target):
        """src/trg: NHWC images as the dataset holds them before RAM -- fundus: uint8 pixels 0..255 (the step is built with
        uint8 RAM buffers; float images are refused by TrainStep.load_raw), prostate: float32 in [-1,1];
        lam: [B]; target: fundus (B,2,H,W) float multilabel mask / prostate (B,H,W) int64."""
        self.ts.load_raw(src_nhwc, trg_nhwc, lam)
        self.ts.load_target(target)
        self._step()

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
