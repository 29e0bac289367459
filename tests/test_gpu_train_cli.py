"""-m gpu: the train.py CLI end to end on a tiny synthetic Fundus tree (PNG files + list files in the
reference's two list locations): 6 iterations, in-training evaluation, checkpoint layout."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_fundus(root, n_per_domain=8, size=64):
    rng = np.random.RandomState(0)
    base = os.path.join(root, 'fundus')
    for d in range(1, 5):
        for split in ('train', 'test'):
            os.makedirs(os.path.join(base, 'Domain%d' % d, split, 'ROIs', 'image'), exist_ok=True)
            os.makedirs(os.path.join(base, 'Domain%d' % d, split, 'ROIs', 'mask'), exist_ok=True)
            lines, partner = [], []
            for i in range(n_per_domain):
                img = rng.randint(0, 255, (size, size, 3)).astype(np.uint8)
                yy, xx = np.mgrid[0:size, 0:size]
                r2 = (yy - size / 2) ** 2 + (xx - size / 2) ** 2
                m = np.full((size, size), 255, np.uint8)
                m[r2 < (size * 0.35) ** 2] = 128
                m[r2 < (size * 0.15) ** 2] = 0
                rel_i = 'Domain%d/%s/ROIs/image/%d.png' % (d, split, i)
                rel_m = 'Domain%d/%s/ROIs/mask/%d.png' % (d, split, i)
                Image.fromarray(img).save(os.path.join(base, rel_i))
                Image.fromarray(m).save(os.path.join(base, rel_m))
                lines.append('%s %s' % (rel_i, rel_m))
                partner.append('%s/ROIs/image/%d.png %s/ROIs/mask/%d.png' % (split, i, split, i))
            with open(os.path.join(base, 'Domain%d_%s.list' % (d, split)), 'w') as f:
                f.write('\n'.join(lines) + '\n')
            if split == 'train':
                with open(os.path.join(base, 'Domain%d' % d, 'train.list'), 'w') as f:
                    f.write('\n'.join(partner) + '\n')
    return root


def test_train_cli_smoke(tmp_path):
    data = _make_fundus(str(tmp_path / 'data'))
    out = str(tmp_path / 'out')
    cmd = [sys.executable, os.path.join(ROOT, 'ram-dsir_amd', 'train.py'), '--data_root', data, '--dataset', 'fundus',
           '--domain_idxs', '1,2,3', '--test_domain_idx', '0', '--ram', '--rec', '--is_out_domain', '--consistency',
           '--consistency_type', 'kd', '--save_path', out, '--epochs', '3', '--max_iters', '6', '--num_workers', '2',
           '--log_every', '2', '--dtype', 'f32']
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    log = r.stdout.decode()
    assert r.returncode == 0, log[-3000:]
    assert 'Encoder Params: 1.968M' in log and 'val_cup_dice' in log
    ck = torch.load(os.path.join(out, 'final_model.pth'), map_location='cpu')
    assert sorted(ck) == ['encoder_state_dict', 'rec_decoder_state_dict', 'seg_decoder_state_dict']
    assert len(ck['encoder_state_dict']) == 105 and len(ck['seg_decoder_state_dict']) == 79 and len(ck['rec_decoder_state_dict']) == 206
    assert int(ck['encoder_state_dict']['convd1.bn1.num_batches_tracked']) == 12          # 6 iterations x 2 passes
    assert all(torch.isfinite(v.float()).all() for v in ck['encoder_state_dict'].values())
    assert os.path.exists(os.path.join(out, '0_val_log.csv'))
    # the reference's tensorboard scalars (train.py:298-304) as an event file in <save_path>/log, at the logged iterations
    sys.path.insert(0, os.path.join(ROOT, 'ram-dsir_amd'))
    from utils import tfevents
    logs = os.listdir(os.path.join(out, 'log'))
    assert len(logs) == 1 and logs[0].startswith('events.out.tfevents.')
    ev = tfevents.read_events(os.path.join(out, 'log', logs[0]))
    assert ev[0]['file_version'] == 'brain.Event:2'
    tags = ['lr', 'loss/loss_bce_1', 'loss/loss_dice_1', 'loss/loss_bce_2', 'loss/loss_dice_2', 'loss/loss_consistency', 'loss/loss_rec']
    assert [(e['step'], e['scalars'][0][0]) for e in ev[1:]] == [(it, t) for it in (0, 2, 4) for t in tags]
    assert all(np.isfinite(e['scalars'][0][1]) for e in ev[1:])
    # offline evaluation script on that checkpoint (BN back in train mode, test_fundus_slice.py:75-83)
    ev = [sys.executable, os.path.join(ROOT, 'ram-dsir_amd', 'test_fundus_slice.py'), '--model_file', os.path.join(out, 'final_model.pth'),
          '--data_dir', data, '--datasetTest', '0', '--test_prediction_save_path', str(tmp_path / 'pred'), '--batch_size', '4']
    r2 = subprocess.run(ev, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r2.returncode == 0, r2.stdout.decode()[-3000:]
    assert 'val_cup_dice' in r2.stdout.decode() and os.path.exists(str(tmp_path / 'pred' / 'test0_log.csv'))


@pytest.mark.parametrize('norm', ['gn', 'in'])
def test_train_cli_with_groupnorm_and_instancenorm(tmp_path, norm):
    """--norm gn / in (code/train.py:69, networks/unet.py:20-23): train.py runs the reference's loop over the drop-in modules instead of
    the fused step; losses fall over 8 iterations on the synthetic tree, the checkpoint has the reference's keys for that norm."""
    data = _make_fundus(str(tmp_path / 'data'))
    out = str(tmp_path / 'out')
    cmd = [sys.executable, os.path.join(ROOT, 'ram-dsir_amd', 'train.py'), '--data_root', data, '--dataset', 'fundus',
           '--domain_idxs', '1,2,3', '--test_domain_idx', '0', '--ram', '--rec', '--is_out_domain', '--consistency',
           '--consistency_type', 'kd', '--save_path', out, '--epochs', '4', '--max_iters', '8', '--num_workers', '2',
           '--log_every', '1', '--dtype', 'f32', '--norm', norm]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    log = r.stdout.decode()
    assert r.returncode == 0, log[-3000:]
    assert 'module-level training loop' in log and 'val_cup_dice' in log
    losses = [float(l.split(' loss ')[1].split()[0]) for l in log.splitlines() if l.startswith('iter ') and ' loss ' in l]
    assert len(losses) == 8 and all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    ck = torch.load(os.path.join(out, 'final_model.pth'), map_location='cpu')
    assert len(ck['encoder_state_dict']) == {'gn': 60, 'in': 30}[norm]
    assert len(ck['seg_decoder_state_dict']) == {'gn': 46, 'in': 24}[norm] and len(ck['rec_decoder_state_dict']) == 206
    assert all(torch.isfinite(v.float()).all() for v in ck['encoder_state_dict'].values())


def test_train_cli_rejects_flag_sets_the_reference_cannot_run(tmp_path):
    cmd = [sys.executable, os.path.join(ROOT, 'ram-dsir_amd', 'train.py'), '--save_path', str(tmp_path), '--epochs', '1', '--ram']
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode != 0 and b'--ram --rec' in r.stdout


def test_prostate_volume_script_on_synthetic_nifti(tmp_path):
    """test_prostate_volume.py end to end: NIfTI volumes of the held-out site, random-init checkpoint, BN in train
    mode; the script must run, report the three metrics and append the reference's CSV line."""
    sys.path.insert(0, os.path.join(ROOT, 'ram-dsir_amd'))
    from networks.unet import Encoder, Decoder
    from utils import nifti
    dom = tmp_path / 'data' / 'prostate' / 'BIDMC'
    os.makedirs(dom)
    rng = np.random.RandomState(3)
    for k in range(2):
        img = rng.uniform(0, 60, (10, 64, 64)).astype(np.float32)
        msk = np.zeros((10, 64, 64), np.uint8)
        img[2:8, 20:44, 20:44] += 150
        msk[2:8, 20:44, 20:44] = 1
        nifti.write_volume(str(dom / ('Case%02d.nii.gz' % k)), img)
        nifti.write_volume(str(dom / ('Case%02d_segmentation.nii.gz' % k)), msk)
    torch.manual_seed(0)
    ck = str(tmp_path / 'ck.pth')
    torch.save({'encoder_state_dict': Encoder().state_dict(), 'seg_decoder_state_dict': Decoder(num_classes=2).state_dict()}, ck)
    ev = [sys.executable, os.path.join(ROOT, 'ram-dsir_amd', 'test_prostate_volume.py'), '--model_file', ck, '--data_dir', str(tmp_path / 'data'),
          '--datasetTest', '4', '--test_prediction_save_path', str(tmp_path / 'pred'), '--batch_size', '4']
    r = subprocess.run(ev, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    log = r.stdout.decode()
    assert r.returncode == 0, log[-3000:]
    assert 'val_dice' in log and 'average_hd' in log and 'average_asd' in log
    line = open(str(tmp_path / 'pred' / 'test4_log.csv')).read()
    assert 'dice coefficence' in line and 'average_asd' in line
