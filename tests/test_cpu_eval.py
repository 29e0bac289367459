"""CPU tests of the evaluation path: the medpy.metric.binary restatement against hand-computed
answers, the NIfTI-1 reader, and the slice batching of the Prostate volume evaluation (code/train.py:134-192,
code/test_prostate_volume.py:79-161) against a straight re-derivation of its rules."""
import gzip
import os
import struct

import numpy as np
import pytest
import torch

from utils import metrics as M
from utils import nifti as N
from utils import prostate_eval as P


# ------------------------------------------------------------------------------------------------ medpy restatement
def test_dc_known_answers():
    a = np.zeros((4, 4), bool); a[:2] = True          # 8 px
    b = np.zeros((4, 4), bool); b[1:3] = True         # 8 px, 4 shared
    assert M.dc(a, b) == pytest.approx(2 * 4 / 16)
    assert M.dc(a, a) == 1.0
    assert M.dc(np.zeros((3, 3)), np.zeros((3, 3))) == 0.0            # medpy: ZeroDivisionError -> 0.0
    assert M.dc(a, np.zeros((4, 4))) == 0.0


def test_surface_metrics_shifted_squares():
    # two 5x5 squares, the second shifted by 3 columns: borders are the square outlines (face erosion)
    a = np.zeros((20, 20), bool); a[5:10, 5:10] = True
    b = np.zeros((20, 20), bool); b[5:10, 8:13] = True
    # identical shapes: zero distance everywhere
    assert M.hd95(a, a) == 0.0 and M.asd(a, a) == 0.0
    # brute-force oracle of the definition: border = x & ~erode4(x); distance to nearest border pixel of the other
    def border(x):
        e = x.copy()
        e[1:] &= x[:-1]; e[:-1] &= x[1:]; e[:, 1:] &= x[:, :-1]; e[:, :-1] &= x[:, 1:]
        e[0] = e[-1] = False; e[:, 0] = e[:, -1] = False
        return x & ~e
    def directed(x, y):
        bx, by = np.argwhere(border(x)), np.argwhere(border(y))
        return np.sqrt(((bx[:, None, :] - by[None, :, :]) ** 2).sum(-1)).min(1)
    d1, d2 = directed(a, b), directed(b, a)
    assert M.asd(a, b) == pytest.approx(d1.mean())
    assert M.asd(b, a) == pytest.approx(d2.mean())
    assert M.hd95(a, b) == pytest.approx(np.percentile(np.hstack((d1, d2)), 95))
    assert M.hd95(a, b) == M.hd95(b, a)
    # voxel spacing scales the distances along that axis
    assert M.asd(a, b, voxelspacing=(1.0, 2.0)) > M.asd(a, b)


def test_surface_metrics_3d_and_empty_operands():
    a = np.zeros((6, 8, 8), bool); a[1:4, 2:6, 2:6] = True
    b = np.zeros((6, 8, 8), bool); b[2:5, 2:6, 2:6] = True
    assert 0.0 < M.asd(a, b) <= 1.0 and M.hd95(a, b) == pytest.approx(1.0)
    with pytest.raises(RuntimeError):
        M.hd95(np.zeros((4, 4)), a[0])
    with pytest.raises(RuntimeError):
        M.asd(a, np.zeros_like(a))


def test_connectivity_region_analysis_keeps_largest_face_connected_component():
    m = np.zeros((3, 6, 6))
    m[0, 0:2, 0:2] = 1                      # 4 voxels
    m[1:3, 3:6, 3:6] = 1                    # 18 voxels
    m[0, 2, 2] = 1                          # touches the first blob only diagonally: its own component
    out = M.connectivity_region_analysis(m)
    assert out.sum() == 18 and out[1:3, 3:6, 3:6].all()
    assert out.dtype.kind == 'i'
    # the reference's quirk on an empty prediction: argmax(sizes) == 0 relabels the background to 1
    assert M.connectivity_region_analysis(np.zeros((2, 3, 3))).sum() == 18


# ------------------------------------------------------------------------------------------------ NIfTI reader
@pytest.mark.parametrize('ext', ['.nii', '.nii.gz'])
@pytest.mark.parametrize('dtype', [np.int16, np.float32, np.uint8])
def test_nifti_round_trip_is_zyx(tmp_path, ext, dtype):
    rng = np.random.RandomState(0)
    vol = (rng.uniform(0, 200, (5, 7, 9))).astype(dtype)               # (z, y, x)
    path = str(tmp_path / ('case' + ext))
    N.write_volume(path, vol)
    got = N.read_volume(path)
    assert got.shape == (5, 7, 9) and got.dtype == np.dtype(dtype)
    np.testing.assert_array_equal(got, vol)
    raw = N._read_bytes(path)
    h = N.read_header(raw)
    assert h['shape'] == (9, 7, 5)                                     # the file stores x fastest
    # first voxels in the file walk along x
    first = np.frombuffer(raw, dtype=h['dtype'], count=9, offset=352)
    np.testing.assert_array_equal(first, vol[0, 0, :])


def test_nifti_big_endian_slope_and_bad_files(tmp_path):
    vol = np.arange(2 * 3 * 4, dtype=np.int16).reshape(2, 3, 4)
    hdr = bytearray(348)
    struct.pack_into('>i', hdr, 0, 348)
    struct.pack_into('>8h', hdr, 40, 3, 4, 3, 2, 1, 1, 1, 1)
    struct.pack_into('>2h', hdr, 70, 4, 16)
    struct.pack_into('>8f', hdr, 76, 1, 1, 1, 1, 1, 1, 1, 1)
    struct.pack_into('>3f', hdr, 108, 352.0, 2.0, -1.0)               # scl_slope 2, scl_inter -1
    hdr[344:348] = b'n+1\x00'
    path = str(tmp_path / 'be.nii.gz')
    with gzip.open(path, 'wb') as f:
        f.write(bytes(hdr) + b'\x00' * 4 + vol.astype('>i2').tobytes())
    got = N.read_volume(path)
    np.testing.assert_allclose(got, vol * 2.0 - 1.0)
    bad = str(tmp_path / 'bad.nii')
    with open(bad, 'wb') as f:
        f.write(b'\x00' * 400)
    with pytest.raises(ValueError):
        N.read_volume(bad)


# ------------------------------------------------------------------------------------------------ volume evaluation
def _threshold_forward(calls):
    """A stand-in network: class 1 where the CENTRE slice of the 2.5-D stack is positive."""
    def fwd(v):
        calls.append(v.clone())
        fg = (v[:, 1:2] > 0).float()
        return torch.cat([1 - fg, fg], 1) * 10
    return fwd


def test_predict_volume_batching_rules():
    rng = np.random.RandomState(1)
    D, Hh, Ww, bs = 11, 6, 6, 4
    image = rng.uniform(0, 100, (D, Hh, Ww))
    image[3:8, 1:5, 1:5] += 300                       # bright block = foreground after min-max normalisation
    mask = np.zeros((D, Hh, Ww), np.int16)
    mask[3:8, 1:5, 1:5] = 2                           # label 2 is folded into 1
    mask[5] = 0                                       # an empty ground-truth slice inside the organ
    calls = []
    post, m = P.predict_volume(_threshold_forward(calls), image, mask, bs)
    assert set(np.unique(m)) == {0, 1}
    # floor(11 / 4) = 2 batches -> frames 1..8 are predicted, frame 9 is not; every batch is 4 wide
    assert len(calls) == 2 and all(c.shape == (bs, 3, Hh, Ww) for c in calls)
    norm = 2 * (image - image.min()) / (image.max() - image.min()) - 1
    np.testing.assert_allclose(calls[0][2].numpy(), norm[2:5], rtol=1e-6)      # slot 2 of batch 0 = frames 2,3,4
    np.testing.assert_allclose(calls[1][3].numpy(), norm[7:10], rtol=1e-6)
    expect = np.zeros((D, Hh, Ww))
    for z in range(1, 9):
        if mask[z].sum() == 0:
            continue                                  # empty ground truth: prediction suppressed
        expect[z] = norm[z] > 0
    # largest face-connected component: slice 5 is empty, so {3,4} and {6,7} are separate blobs of equal size -> first wins
    lab = M.connectivity_region_analysis(expect)
    np.testing.assert_array_equal(post, lab)
    assert post[5].sum() == 0 and post[9].sum() == 0


def test_predict_volume_short_tail_batch_is_zero_padded():
    D, bs = 9, 4                                      # frames 1..7; batch 1 holds frames 5,6,7 + one all-zero slot
    image = np.linspace(0, 1, D * 4 * 4).reshape(D, 4, 4)
    mask = np.ones((D, 4, 4), np.uint8)
    calls = []
    P.predict_volume(_threshold_forward(calls), image, mask, bs)
    assert len(calls) == 2
    assert float(calls[1][3].abs().sum()) == 0.0 and float(calls[1][2].abs().sum()) > 0.0


def test_evaluate_domain_reads_nifti_pairs(tmp_path):
    dom = tmp_path / 'prostate' / 'BIDMC'
    os.makedirs(dom)
    rng = np.random.RandomState(2)
    for k in range(2):
        img = rng.uniform(0, 50, (8, 6, 6)).astype(np.float32)
        msk = np.zeros((8, 6, 6), np.uint8)
        img[2:6, 1:4, 1:4] += 200
        msk[2:6, 1:4, 1:4] = 1
        N.write_volume(str(dom / ('Case%02d.nii.gz' % k)), img)
        N.write_volume(str(dom / ('Case%02d_segmentation.nii.gz' % k)), msk)
    assert sorted(P.volume_files(str(tmp_path / 'prostate'), 'BIDMC')) == ['Case00.nii.gz', 'Case01.nii.gz']
    dice, hd, sd = P.evaluate_domain(_threshold_forward([]), str(tmp_path / 'prostate'), 'BIDMC', 2, with_surface=True)
    assert dice == pytest.approx(1.0) and hd == 0.0 and sd == 0.0
    assert P.DOMAIN_LIST[4] == 'BIDMC' and len(P.DOMAIN_LIST) == 6


# ------------------------------------------------------------------------------------------------ NIfTI-1, byte level
def _nifti1_header(endian, dims_xyz, datatype, bitpix, vox_offset, slope, inter, pixdim=(1.0, 0.5, 0.5, 3.0)):
    """348 header bytes laid out field by field from the published nifti1.h (NOT through utils/nifti.write_volume):
    sizeof_hdr int32 @0 | dim[8] int16 @40 | datatype int16 @70 | bitpix int16 @72 | pixdim[8] float32 @76 |
    vox_offset float32 @108 | scl_slope float32 @112 | scl_inter float32 @116 | qform_code / sform_code int16 @252/254 |
    magic char[4] @344."""
    h = bytearray(348)
    struct.pack_into(endian + 'i', h, 0, 348)
    struct.pack_into(endian + '8h', h, 40, len(dims_xyz), *(list(dims_xyz) + [1] * (7 - len(dims_xyz))))
    struct.pack_into(endian + 'h', h, 70, datatype)
    struct.pack_into(endian + 'h', h, 72, bitpix)
    struct.pack_into(endian + '8f', h, 76, *(list(pixdim) + [0.0] * (8 - len(pixdim))))
    struct.pack_into(endian + 'f', h, 108, float(vox_offset))
    struct.pack_into(endian + 'f', h, 112, slope)
    struct.pack_into(endian + 'f', h, 116, inter)
    struct.pack_into(endian + '2h', h, 252, 1, 1)
    h[344:348] = b'n+1\x00'
    return bytes(h)


def test_nifti_reader_on_hand_built_files(tmp_path):
    """Files assembled byte by byte, as a scanner export / another library would write them: a big-endian int16 volume with
    scl_slope / scl_inter set and a header extension in front of the voxels (vox_offset 368), gzip-compressed; and a plain
    little-endian uint8 label volume.  Expected arrays follow from the format alone: x runs fastest in the file, and
    sitk.GetArrayFromImage hands numpy (z, y, x)."""
    nx, ny, nz = 5, 4, 3
    vox = np.arange(nx * ny * nz, dtype=np.int16) * 7 - 100              # file order: x fastest, then y, then z
    hdr = _nifti1_header('>', (nx, ny, nz), 4, 16, 368, 2.0, -1.0)
    ext = bytes([1, 0, 0, 0]) + struct.pack('>2i', 16, 4) + b'comment\x00'   # extender + one 16-byte extension (esize, ecode, 8 data bytes)
    assert len(hdr) + len(ext) == 368
    p = str(tmp_path / 'be.nii.gz')
    with gzip.open(p, 'wb') as f:
        f.write(hdr + ext + vox.astype('>i2').tobytes())
    arr = N.read_volume(p)
    assert arr.shape == (nz, ny, nx)
    want = vox.astype(np.float64).reshape(nz, ny, nx) * 2.0 - 1.0       # ITK applies slope / intercept
    np.testing.assert_array_equal(arr, want)
    assert arr[1, 2, 3] == (3 + nx * (2 + ny * 1)) * 7 * 2.0 - 100 * 2.0 - 1.0
    h = N.read_header(gzip.open(p, 'rb').read())
    assert h['endian'] == '>' and h['shape'] == (nx, ny, nz) and h['vox_offset'] == 368 and tuple(h['pixdim']) == (0.5, 0.5, 3.0)

    lab = (np.arange(nx * ny * nz) % 3).astype(np.uint8)
    p2 = str(tmp_path / 'le.nii')
    with open(p2, 'wb') as f:
        f.write(_nifti1_header('<', (nx, ny, nz), 2, 8, 352, 0.0, 0.0) + bytes(4) + lab.tobytes())      # slope 0: "do not scale"
    arr2 = N.read_volume(p2)
    assert arr2.dtype == np.uint8 and arr2.shape == (nz, ny, nx)
    np.testing.assert_array_equal(arr2, lab.reshape(nz, ny, nx))
    # 4-D file with a trailing singleton time axis (common in exports): squeezed like SimpleITK's 3-D read of it
    p3 = str(tmp_path / 'le4d.nii')
    with open(p3, 'wb') as f:
        f.write(_nifti1_header('<', (nx, ny, nz, 1), 16, 32, 352, 1.0, 0.0) + bytes(4) + np.arange(60, dtype='<f4').tobytes())
    arr3 = N.read_volume(p3)
    assert arr3.shape == (nz, ny, nx) and arr3.dtype == np.float32 and arr3[2, 3, 4] == 59.0
    # and the repo's own writer produces a header this layout reads back (cross-check of write_volume against the field table)
    p4 = str(tmp_path / 'w.nii')
    N.write_volume(p4, lab.reshape(nz, ny, nx), pixdim=(0.5, 0.5, 3.0))
    raw = open(p4, 'rb').read()
    assert struct.unpack('<i', raw[0:4])[0] == 348 and struct.unpack('<8h', raw[40:56])[:4] == (3, nx, ny, nz)
    assert struct.unpack('<2h', raw[70:74]) == (2, 8) and struct.unpack('<f', raw[108:112])[0] == 352.0 and raw[344:348] == b'n+1\x00'
    assert raw[352:] == lab.tobytes()
