"""-m gpu: NUMERIC parity of the two evaluation pipelines on the GPU box.

The product scripts (ram-dsir_amd/test_fundus_slice.py, ram-dsir_amd/test_prostate_volume.py -- HIP forward) run on a synthetic
tree with a fixed checkpoint; the SAME pipeline is then driven by the ORACLE forward on the CPU (oracle.unet, BatchNorm in train
mode = batch statistics of each test batch, batches in list order: code/test_fundus_slice.py:75-83,104-137;
code/test_prostate_volume.py:65-74,100-118) and the reported numbers are compared: Dice, HD95, ASD.

What this pins: checkpoint loading, BN-train-mode inference through the fused forward, sigmoid / softmax + argmax, the bilinear
resize to the native mask size, the 0.75 threshold, connected components / hole filling, 2.5-D stacking with the reference's
floor(D / batch) batching and zero-padded tail slots, empty-ground-truth suppression, 3-D largest component -- end to end, on
the numbers the reference prints.  The metric FUNCTIONS are pinned separately (tests/test_cpu_sampling.py: Dice against the
reference's own output; tests/test_cpu_eval.py: hd95 / asd / post-processing against hand-derived answers).
Tolerances: the fp32 HIP forward differs from the oracle's by ~1e-5 relative, which can move a probability across the
threshold on a handful of border pixels: Dice within 2e-3 absolute, the surface distances within a fraction of a pixel."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'ram-dsir_amd')


def _states(sharpen):
    """A fixed random checkpoint whose logits are spread enough to give structured masks: the output conv is scaled (a
    random-init network answers sigmoid ~ 0.5 everywhere, which the 0.75 threshold turns into empty masks)."""
    from oracle import unet as OU
    enc, dec = OU.encoder_state(seed=11), OU.decoder_state(num_classes=2, seed=12)
    dec['out1.weight'] = dec['out1.weight'] * sharpen
    return enc, dec


def _save_ck(path, enc, dec):
    torch.save({'encoder_state_dict': {k: v.clone() for k, v in enc.items()},
                'seg_decoder_state_dict': {k: v.clone() for k, v in dec.items()}}, path)


def _csv_floats(path):
    """The numbers behind the labels of the reference's CSV line (test_fundus_slice.py:163-170)."""
    toks = open(path).read().strip().splitlines()[-1].replace('[', '').replace(']', '').split(',')
    vals, out = [t.strip().strip("'") for t in toks], {}
    for i, t in enumerate(vals):
        if t.endswith(': ') or t.endswith(':'):
            try:
                out[t.strip(': ').strip()] = float(vals[i + 1])
            except ValueError:
                pass
    return out


def test_fundus_slice_script_matches_the_oracle_driven_pipeline(tmp_path):
    import synth_data
    sys.path.insert(0, PKG)
    import dataset.transform as trans
    from dataset.fundus import Fundus
    from utils.metrics import asd, dice_coeff_2label, hd95, postprocessing
    from oracle import unet as OU

    data = str(tmp_path / 'data')
    synth_data.make_fundus_tree(data, n_train=1, n_test=10, hw=(136, 152), vary=False)
    enc, dec = _states(sharpen=10.0)
    ck = str(tmp_path / 'ck.pth')
    _save_ck(ck, enc, dec)
    bs = 8                                                            # 10 test images: a batch of 8 and a batch of 2
    cmd = [sys.executable, os.path.join(PKG, 'test_fundus_slice.py'), '--model_file', ck, '--data_dir', data, '--datasetTest', '0',
           '--test_prediction_save_path', str(tmp_path / 'pred'), '--batch_size', str(bs)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    got = _csv_floats(str(tmp_path / 'pred' / 'test0_log.csv'))

    # ---- the same pipeline with the oracle forward (CPU, fp32, BatchNorm in train mode)
    class Compose(object):
        def __init__(self, ts): self.ts = ts
        def __call__(self, s):
            for t in self.ts:
                s = t(s)
            return s
    ds = Fundus(base_dir=os.path.join(data, 'fundus'), split='test', domain_idx=0, transform=Compose([trans.Resize((256, 256)), trans.Normalize()]))
    cup = disc = 0.0
    hd, sd, n, fg = [0.0, 0.0], [0.0, 0.0], 0, []
    with torch.no_grad():
        for b0 in range(0, len(ds), bs):
            items = [ds[i] for i in range(b0, min(b0 + bs, len(ds)))]
            x = torch.stack([it[0] for it in items])
            torig = torch.stack([it[2] for it in items])
            e2, d2 = OU.clone_state(enc), OU.clone_state(dec)          # train-mode BN writes running statistics: keep the checkpoint
            pred = torch.sigmoid(OU.decoder_forward(OU.encoder_forward(x, e2, True), d2, True))
            pred = F.interpolate(pred, size=(torig.size(2), torig.size(3)), mode='bilinear')
            for i in range(pred.shape[0]):
                post = postprocessing(pred[i], dataset='fundus', threshold=0.75)
                c, d = dice_coeff_2label(post, torig[i])
                cup, disc, n = cup + c, disc + d, n + 1
                fg.append(float(post.mean()))
                for k in (0, 1):
                    if np.sum(post[k]) < 1e-4:
                        hd[k] += 100
                        sd[k] += 100
                    else:
                        hd[k] += hd95(post[k].astype(bool), torig[i, k].numpy().astype(bool))
                        sd[k] += asd(post[k].astype(bool), torig[i, k].numpy().astype(bool))
    want = {'cup dice coefficence': cup / n, 'disc dice coefficence': disc / n, 'average_hd_OC': hd[0] / n, 'average_hd_OD': hd[1] / n,
            'average_asd_OC': sd[0] / n, 'average_asd_OD': sd[1] / n}
    # the checkpoint really produces structured, non-trivial masks (otherwise every number below is a constant)
    assert 0.02 < float(np.mean(fg)) < 0.9 and np.std(fg) > 0, fg
    assert want['average_hd_OD'] < 100 and want['average_hd_OC'] < 100
    for k in ('cup dice coefficence', 'disc dice coefficence'):
        assert abs(got[k] - want[k]) < 2e-3, (k, got[k], want[k])
    for k in ('average_hd_OC', 'average_hd_OD'):
        assert abs(got[k] - want[k]) < 0.5, (k, got[k], want[k])         # one border pixel of one image moves hd95 by <= 1 / n
    for k in ('average_asd_OC', 'average_asd_OD'):
        assert abs(got[k] - want[k]) < 0.05, (k, got[k], want[k])


def test_prostate_volume_script_matches_the_oracle_driven_pipeline(tmp_path):
    sys.path.insert(0, PKG)
    from utils import nifti
    from utils.prostate_eval import evaluate_domain
    from oracle import unet as OU

    dom = tmp_path / 'data' / 'prostate' / 'BIDMC'
    os.makedirs(dom)
    rng = np.random.RandomState(5)
    for k in range(2):
        D = 11 + k                                                    # 11 and 12 slices: batches of 4 -> floor(D/4) = 2 / 3 batches, a ragged tail
        img = rng.uniform(0, 60, (D, 64, 64)).astype(np.float32)
        msk = np.zeros((D, 64, 64), np.uint8)
        img[2:8, 18:46, 20:44] += 150 + 20 * k
        msk[2:8, 18:46, 20:44] = 1 + (k % 2)                          # label 2 is folded into 1 (test_prostate_volume.py:95)
        nifti.write_volume(str(dom / ('Case%02d.nii.gz' % k)), img)
        nifti.write_volume(str(dom / ('Case%02d_segmentation.nii.gz' % k)), msk)
    enc, dec = _states(sharpen=25.0)
    ck = str(tmp_path / 'ck.pth')
    _save_ck(ck, enc, dec)
    cmd = [sys.executable, os.path.join(PKG, 'test_prostate_volume.py'), '--model_file', ck, '--data_dir', str(tmp_path / 'data'),
           '--datasetTest', '4', '--test_prediction_save_path', str(tmp_path / 'pred'), '--batch_size', '4']
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    got = _csv_floats(str(tmp_path / 'pred' / 'test4_log.csv'))

    def oracle_forward(v):
        e2, d2 = OU.clone_state(enc), OU.clone_state(dec)
        with torch.no_grad():
            return OU.decoder_forward(OU.encoder_forward(v, e2, True), d2, True)
    # same file order as the script (os.listdir inside evaluate_domain)
    dice, hd, sd = evaluate_domain(oracle_forward, str(tmp_path / 'data' / 'prostate'), 'BIDMC', 4, with_surface=True)
    assert 0.0 < dice < 1.0, dice                                      # a structured, imperfect prediction
    assert abs(got['dice coefficence'] - dice) < 2e-3, (got, dice)
    assert abs(got['average_hd'] - hd) < 0.5, (got, hd)
    assert abs(got['average_asd'] - sd) < 0.05, (got, sd)
