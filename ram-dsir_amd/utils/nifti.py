"""Minimal NIfTI-1 reader (single-file .nii / .nii.gz) in numpy.

The reference reads the Prostate volumes with ``sitk.GetArrayFromImage(sitk.ReadImage(path))``
(code/train.py:147-151, code/test_prostate_volume.py:87-91); SimpleITK is not installed in this image and cannot be
vendored, so the on-disk format is read directly.  What the call pair returns, and what ``read_volume`` returns:
the voxel array in FILE order with the axes reversed -- NIfTI stores x fastest, SimpleITK hands numpy (z, y, x) --
with ``scl_slope``/``scl_inter`` applied when the header sets them (ITK's NiftiImageIO rescales and then reports a
floating pixel type), and no re-orientation (ITK keeps the voxel grid and puts qform/sform into the direction matrix).
Layout follows the published NIfTI-1 header (348 bytes; nifti1.h)."""
import gzip
import struct

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32, 1024: np.int64, 1280: np.uint64}


def _read_bytes(path):
    with open(path, 'rb') as f:
        head = f.read(2)
    if head == b'\x1f\x8b':
        with gzip.open(path, 'rb') as f:
            return f.read()
    with open(path, 'rb') as f:
        return f.read()


def read_header(raw):
    if len(raw) < 348:
        raise ValueError('not a NIfTI-1 file: shorter than the 348-byte header')
    for endian in ('<', '>'):
        if struct.unpack(endian + 'i', raw[0:4])[0] == 348:
            break
    else:
        raise ValueError('not a NIfTI-1 file: sizeof_hdr != 348')
    magic = raw[344:348]
    if magic not in (b'n+1\x00', b'ni1\x00'):
        raise ValueError('not a NIfTI-1 file: bad magic %r' % magic)
    if magic == b'ni1\x00':
        raise ValueError('two-file NIfTI (.hdr/.img) is not supported')
    dim = struct.unpack(endian + '8h', raw[40:56])
    datatype, bitpix = struct.unpack(endian + '2h', raw[70:74])
    pixdim = struct.unpack(endian + '8f', raw[76:108])
    vox_offset, slope, inter = struct.unpack(endian + '3f', raw[108:120])
    if datatype not in _DTYPES:
        raise ValueError('unsupported NIfTI datatype code %d' % datatype)
    ndim = dim[0]
    if not 1 <= ndim <= 7:
        raise ValueError('bad NIfTI dim[0] = %d' % ndim)
    return dict(endian=endian, shape=tuple(int(d) for d in dim[1:1 + ndim]), dtype=np.dtype(_DTYPES[datatype]).newbyteorder(endian),
                pixdim=pixdim[1:1 + ndim], vox_offset=int(vox_offset), slope=float(slope), inter=float(inter))


def read_volume(path):
    """(z, y, x) array as sitk.GetArrayFromImage(sitk.ReadImage(path)) gives it (trailing singleton dims dropped)."""
    raw = _read_bytes(path)
    h = read_header(raw)
    shape = h['shape']
    while len(shape) > 3 and shape[-1] == 1:
        shape = shape[:-1]
    n = int(np.prod(shape))
    off = max(h['vox_offset'], 352)
    data = np.frombuffer(raw, dtype=h['dtype'], count=n, offset=off)
    arr = data.reshape(shape[::-1]).astype(h['dtype'].newbyteorder('='))          # x fastest in the file -> (…, z, y, x)
    slope, inter = h['slope'], h['inter']
    if slope != 0.0 and not (slope == 1.0 and inter == 0.0) and np.isfinite(slope) and np.isfinite(inter):
        arr = arr.astype(np.float64) * slope + inter
    return arr


def write_volume(path, arr_zyx, pixdim=(1.0, 1.0, 1.0)):
    """Inverse of read_volume for tests / exports: (z, y, x) array -> single-file NIfTI-1 (.nii or .nii.gz)."""
    arr = np.ascontiguousarray(arr_zyx)
    code = [k for k, v in _DTYPES.items() if np.dtype(v) == arr.dtype]
    if not code:
        raise ValueError('unsupported dtype %s' % arr.dtype)
    shape = arr.shape[::-1]
    hdr = bytearray(348)
    struct.pack_into('<i', hdr, 0, 348)
    struct.pack_into('<8h', hdr, 40, len(shape), *(list(shape) + [1] * (7 - len(shape))))
    struct.pack_into('<2h', hdr, 70, code[0], arr.dtype.itemsize * 8)
    struct.pack_into('<8f', hdr, 76, 1.0, *(list(pixdim) + [1.0] * (7 - len(pixdim))))
    struct.pack_into('<3f', hdr, 108, 352.0, 1.0, 0.0)
    hdr[344:348] = b'n+1\x00'
    blob = bytes(hdr) + b'\x00' * 4 + arr.astype(arr.dtype.newbyteorder('<')).tobytes()
    if str(path).endswith('.gz'):
        with gzip.open(path, 'wb') as f:
            f.write(blob)
    else:
        with open(path, 'wb') as f:
            f.write(blob)
