#!/usr/bin/env python3
"""Drop-in for the reference's code/test_fundus_slice.py (offline Fundus evaluation, :46-175): load a checkpoint,
run Encoder + Decoder on the held-out domain with -- unless --freeze_bn -- every BatchNorm2d back in train mode
(batch statistics of each test batch, :75-83), resize to the native mask size, threshold 0.75, largest connected
component + hole filling, Dice with +1 smoothing, HD95 / ASD per structure (:110-136; 100 when the post-processed
prediction is empty) through utils/metrics.py's restatement of medpy.metric.binary (medpy is absent here)."""
import argparse
import os
import os.path as osp
import sys

HERE = osp.dirname(osp.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import torch
import torch.nn as nn
from torch.utils.data import DataLoader

import dataset.transform as trans
from dataset.fundus import Fundus
from networks.unet import Encoder, Decoder
import numpy as np

from utils.metrics import asd, dice_coeff_2label, hd95, postprocessing
from train import Compose


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--model_file', type=str, required=True)
    p.add_argument('--dataset', type=str, default='fundus')
    p.add_argument('--data_dir', default='../dataset')
    p.add_argument('--datasetTest', type=int, default=3)
    p.add_argument('--in_channels', type=int, default=3)
    p.add_argument('--batch_size', type=int, default=8)
    p.add_argument('--num_classes', type=int, default=2)
    p.add_argument('--test_prediction_save_path', type=str, required=True)
    p.add_argument('--save_result', action='store_true')
    p.add_argument('--freeze_bn', action='store_true')
    p.add_argument('--norm', type=str, default='bn')
    p.add_argument('--activation', type=str, default='relu')
    p.add_argument('--gpu', type=str, default='0')
    return p.parse_args(argv)


def main(args):
    data_dir = os.path.join(args.data_dir, args.dataset)
    os.makedirs(args.test_prediction_save_path, exist_ok=True)
    testset = Fundus(base_dir=data_dir, split='test', domain_idx=args.datasetTest,
                     transform=Compose([trans.Resize((256, 256)), trans.Normalize()]))
    loader = DataLoader(testset, batch_size=args.batch_size, shuffle=False, num_workers=2, drop_last=False)
    encoder = Encoder(c=args.in_channels, norm=args.norm, activation=args.activation).cuda()
    seg_decoder = Decoder(num_classes=args.num_classes, norm=args.norm, activation=args.activation).cuda()
    ck = torch.load(args.model_file, map_location='cpu')
    encoder.load_state_dict(ck['encoder_state_dict'])
    seg_decoder.load_state_dict(ck['seg_decoder_state_dict'])
    encoder.eval()
    seg_decoder.eval()
    if not args.freeze_bn:                                             # test_fundus_slice.py:75-83
        for m in list(encoder.modules()) + list(seg_decoder.modules()):
            if isinstance(m, nn.BatchNorm2d):
                m.train()
    cup = disc = 0.0
    hd = [0.0, 0.0]
    sd = [0.0, 0.0]
    n = 0
    with torch.no_grad():
        for data, target, target_orig, ids in loader:
            pred = torch.sigmoid(seg_decoder(encoder(data.cuda())))
            pred = torch.nn.functional.interpolate(pred, size=(target_orig.size(2), target_orig.size(3)), mode='bilinear')
            tnp = target_orig.numpy()
            for i in range(pred.shape[0]):
                post = postprocessing(pred[i], dataset=args.dataset, threshold=0.75)
                c, d = dice_coeff_2label(post, target_orig[i])
                cup, disc, n = cup + c, disc + d, n + 1
                for k in (0, 1):                                   # 0 = cup (OC), 1 = disc (OD); :115-136
                    if np.sum(post[k]) < 1e-4:
                        hd[k] += 100
                        sd[k] += 100
                    else:
                        hd[k] += hd95(post[k].astype(bool), tnp[i, k].astype(bool))
                        sd[k] += asd(post[k].astype(bool), tnp[i, k].astype(bool))
    m = max(n, 1)
    cup, disc = cup / m, disc / m
    hd = [v / m for v in hd]
    sd = [v / m for v in sd]
    print('''\n==>val_cup_dice : %.2f''' % (100 * cup))
    print('''\n==>val_disc_dice : %.2f''' % (100 * disc))
    print('''\n==>average_hd_OC : %.2f''' % hd[0])
    print('''\n==>average_hd_OD : %.2f''' % hd[1])
    print('''\n==>average_asd_OC : %.2f''' % sd[0])
    print('''\n==>average_asd_OD : %.2f''' % sd[1])
    with open(osp.join(args.test_prediction_save_path, 'test' + str(args.datasetTest) + '_log.csv'), 'a') as f:
        log = [['batch-size: '] + [args.batch_size] + [args.model_file] + ['cup dice coefficence: '] + [cup] +
               ['disc dice coefficence: '] + [disc] + ['average_hd_OC: '] + [hd[0]] + ['average_hd_OD: '] + [hd[1]] +
               ['average_asd_OC: '] + [sd[0]] + ['average_asd_OD: '] + [sd[1]]]
        f.write(','.join(map(str, log)) + '\n')
    return cup, disc


if __name__ == '__main__':
    a = parse_args()
    os.environ['CUDA_VISIBLE_DEVICES'] = a.gpu
    main(a)
