#!/usr/bin/env python3
"""Drop-in for the reference's code/test_prostate_volume.py (offline Prostate evaluation on 3-D volumes, :40-161):
load a checkpoint, run Encoder + Decoder over the NIfTI volumes of the held-out site with -- unless --freeze_bn --
every BatchNorm2d back in train mode (batch statistics of each slice batch, :65-74), 2.5-D three-slice inputs, argmax,
empty-ground-truth slices suppressed, largest 3-D connected component, then Dice / HD95 / ASD per volume and their
means (:121-159).  NIfTI is read by utils/nifti.py and the three metrics are utils/metrics.py's restatement of
medpy.metric.binary (neither SimpleITK nor medpy exists in this image).  --save_result (bmp overlays, :129-142) is
accepted and ignored."""
import argparse
import os
import os.path as osp
import sys

HERE = osp.dirname(osp.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import torch
import torch.nn as nn

from networks.unet import Encoder, Decoder
from utils.prostate_eval import DOMAIN_LIST, evaluate_domain


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='Test on Prostate dataset (3D volume)')
    p.add_argument('--model_file', type=str, default=None, required=True, help='Model path')
    p.add_argument('--dataset', type=str, default='prostate', help='training dataset')
    p.add_argument('--data_dir', default='../dataset', help='data root path')
    p.add_argument('--datasetTest', type=int, default=3, help='test folder id contain images ROIs to test')
    p.add_argument('--in_channels', type=int, default=3, help='number of input channels')
    p.add_argument('--batch_size', type=int, default=8, help='batch size of testing')
    p.add_argument('--num_classes', type=int, default=2, help='number of classes')
    p.add_argument('--test_prediction_save_path', type=str, default=None, required=True, help='Path root for test image and mask')
    p.add_argument('--save_result', action='store_true', help='Save Results')
    p.add_argument('--freeze_bn', action='store_true', help='Freeze Batch Normalization')
    p.add_argument('--norm', type=str, default='bn', help='normalization type')
    p.add_argument('--activation', type=str, default='relu', help='feature activation function')
    p.add_argument('--gpu', type=str, default='0', help='GPU to use')
    return p.parse_args(argv)


def main(args):
    domain_name = DOMAIN_LIST[args.datasetTest]
    data_dir = os.path.join(args.data_dir, args.dataset)
    os.makedirs(args.test_prediction_save_path, exist_ok=True)
    output_path = os.path.join(args.test_prediction_save_path, 'test' + str(args.datasetTest))
    os.makedirs(output_path, exist_ok=True)

    encoder = Encoder(c=args.in_channels, norm=args.norm, activation=args.activation).cuda()
    seg_decoder = Decoder(num_classes=args.num_classes, norm=args.norm, activation=args.activation).cuda()
    ck = torch.load(args.model_file, map_location='cpu')
    encoder.load_state_dict(ck['encoder_state_dict'])
    seg_decoder.load_state_dict(ck['seg_decoder_state_dict'])
    encoder.eval()
    seg_decoder.eval()
    if not args.freeze_bn:                                             # test_prostate_volume.py:65-74
        for m in list(encoder.modules()) + list(seg_decoder.modules()):
            if isinstance(m, nn.BatchNorm2d):
                m.train()
    with torch.no_grad():
        val_dice, total_hd, total_asd = evaluate_domain(lambda v: seg_decoder(encoder(v.cuda())), data_dir, domain_name,
                                                        args.batch_size, with_surface=True)
    print('''\n==>val_dice : %.2f''' % (100 * val_dice))
    print('''\n==>average_hd : %.2f''' % total_hd)
    print('''\n==>average_asd : %.2f''' % total_asd)
    with open(osp.join(output_path, '../test' + str(args.datasetTest) + '_log.csv'), 'a') as f:
        log = [['batch-size: '] + [args.batch_size] + [args.model_file] + ['dice coefficence: '] + [val_dice] +
               ['average_hd: '] + [total_hd] + ['average_asd: '] + [total_asd]]
        f.write(','.join(map(str, log)) + '\n')
    return val_dice, total_hd, total_asd


if __name__ == '__main__':
    a = parse_args()
    os.environ['CUDA_VISIBLE_DEVICES'] = a.gpu
    main(a)
