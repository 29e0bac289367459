"""Minimal TensorBoard event-file writer for scalars: what the reference's `tensorboardX.SummaryWriter(save_path + '/log')`
produces for `writer.add_scalar(tag, value, iter_num)` (code/train.py:298-304, 467-473, 538).  tensorboardX / tensorboard are
not installed here, so the three pieces are written out by hand:

  * TFRecord framing: uint64 length | masked crc32c(length) | payload | masked crc32c(payload), little endian;
  * crc32c (Castagnoli, reflected polynomial 0x82F63B78) with TensorFlow's mask ((crc >> 15 | crc << 17) + 0xa282ead8);
  * protobuf wire format of  Event { double wall_time = 1; int64 step = 2; string file_version = 3; Summary summary = 5; }
    Summary { repeated Value value = 1; }   Value { string tag = 1; float simple_value = 2; }.

The first record of a file is Event{wall_time, file_version: "brain.Event:2"}; the file name follows TensorBoard's
`events.out.tfevents.<unix time>.<hostname>` pattern so that `tensorboard --logdir <save_path>/log` picks it up.
Image summaries (train.py:306-329) are out of scope.
"""
import os
import socket
import struct
import time

_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _TABLE.append(_c)


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    n &= (1 << 64) - 1                     # int64 fields: two's complement, ten bytes when negative
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field, wire):
    return _varint((field << 3) | wire)


def _bytes_field(field, payload):
    return _key(field, 2) + _varint(len(payload)) + payload


def encode_event(wall_time, step=None, file_version=None, scalars=None):
    """scalars: list of (tag, float)."""
    ev = _key(1, 1) + struct.pack('<d', wall_time)
    if step is not None:
        ev += _key(2, 0) + _varint(int(step))
    if file_version is not None:
        ev += _bytes_field(3, file_version.encode())
    if scalars:
        summ = b''
        for tag, v in scalars:
            val = _bytes_field(1, tag.encode()) + _key(2, 5) + struct.pack('<f', float(v))
            summ += _bytes_field(1, val)
        ev += _bytes_field(5, summ)
    return ev


def frame(payload):
    head = struct.pack('<Q', len(payload))
    return head + struct.pack('<I', masked_crc(head)) + payload + struct.pack('<I', masked_crc(payload))


class SummaryWriter(object):
    """`add_scalar(tag, scalar_value, global_step)` / `flush()` / `close()` of tensorboardX.SummaryWriter."""

    def __init__(self, logdir, flush_secs=30):
        os.makedirs(logdir, exist_ok=True)
        self.logdir = logdir
        self.path = os.path.join(logdir, 'events.out.tfevents.%010d.%s' % (int(time.time()), socket.gethostname()))
        self._f = open(self.path, 'wb')
        self._f.write(frame(encode_event(time.time(), file_version='brain.Event:2')))
        self._flush_secs, self._last = flush_secs, time.time()

    def add_scalar(self, tag, scalar_value, global_step=None, walltime=None):
        self.add_scalars_at(global_step, [(tag, scalar_value)], walltime)

    def add_scalars_at(self, global_step, pairs, walltime=None):
        """Several scalars of one step, one record each (as consecutive add_scalar calls write them)."""
        t = time.time() if walltime is None else walltime
        for tag, v in pairs:
            self._f.write(frame(encode_event(t, step=global_step, scalars=[(tag, v)])))
        if t - self._last > self._flush_secs:
            self.flush()

    def flush(self):
        self._f.flush()
        self._last = time.time()

    def close(self):
        if not self._f.closed:
            self._f.flush()
            self._f.close()


# ---------------------------------------------------------------------------------------------- reader (tests, tooling)
def _read_varint(buf, pos):
    shift = n = 0
    while True:
        b = buf[pos]
        pos += 1
        n |= (b & 0x7F) << shift
        if not b & 0x80:
            return n, pos
        shift += 7


def _fields(buf):
    pos = 0
    while pos < len(buf):
        k, pos = _read_varint(buf, pos)
        field, wire = k >> 3, k & 7
        if wire == 0:
            v, pos = _read_varint(buf, pos)
        elif wire == 1:
            v, pos = buf[pos:pos + 8], pos + 8
        elif wire == 5:
            v, pos = buf[pos:pos + 4], pos + 4
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            v, pos = buf[pos:pos + n], pos + n
        else:
            raise ValueError('wire type %d' % wire)
        yield field, wire, v


def read_events(path):
    """-> list of dicts {wall_time, step, file_version, scalars: [(tag, value)]}; every CRC is verified."""
    out = []
    with open(path, 'rb') as f:
        data = f.read()
    pos = 0
    while pos < len(data):
        head = data[pos:pos + 8]
        (n,) = struct.unpack('<Q', head)
        (c1,) = struct.unpack('<I', data[pos + 8:pos + 12])
        payload = data[pos + 12:pos + 12 + n]
        (c2,) = struct.unpack('<I', data[pos + 12 + n:pos + 16 + n])
        if c1 != masked_crc(head) or c2 != masked_crc(payload) or len(payload) != n:
            raise ValueError('corrupt record at byte %d' % pos)
        pos += 16 + n
        ev = dict(wall_time=None, step=0, file_version=None, scalars=[])
        for field, wire, v in _fields(payload):
            if field == 1:
                ev['wall_time'] = struct.unpack('<d', v)[0]
            elif field == 2:
                ev['step'] = v - (1 << 64) if v >> 63 else v
            elif field == 3:
                ev['file_version'] = v.decode()
            elif field == 5:
                for f1, _, val in _fields(v):
                    if f1 != 1:
                        continue
                    tag, sv = None, None
                    for f2, _, x in _fields(val):
                        if f2 == 1:
                            tag = x.decode()
                        elif f2 == 2:
                            sv = struct.unpack('<f', x)[0]
                    ev['scalars'].append((tag, sv))
        out.append(ev)
    return out
