"""utils/tfevents.py: the TensorBoard scalar event files behind the reference's writer.add_scalar calls (code/train.py:298-304).
tensorboard / tensorboardX are not installed, so the format is pinned by known answers: CRC-32C check values (RFC 3720
B.4), TensorFlow's CRC mask, and the protobuf bytes of one Event written out by hand."""
import os
import struct

import pytest

from utils import tfevents as tfe


def test_crc32c_known_answers():
    assert tfe.crc32c(b'123456789') == 0xE3069283                       # the CRC catalogue's check value for CRC-32C
    assert tfe.crc32c(bytes(32)) == 0x8A9136AA                          # RFC 3720 B.4: 32 bytes of zeros
    assert tfe.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43                 # RFC 3720 B.4: 32 bytes of ones
    assert tfe.crc32c(bytes(range(32))) == 0x46DD794E                   # RFC 3720 B.4: incrementing bytes
    c = tfe.crc32c(b'123456789')
    assert tfe.masked_crc(b'123456789') == ((((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF)


def test_event_protobuf_bytes_by_hand():
    # Event{wall_time = 1.5 (field 1, 64-bit), step = 3 (field 2, varint), summary (field 5) {value (1) {tag (1) "lr", simple_value (2, 32-bit) 0.5}}}
    value = bytes([0x0A, 0x02]) + b'lr' + bytes([0x15]) + struct.pack('<f', 0.5)
    summary = bytes([0x0A, len(value)]) + value
    want = bytes([0x09]) + struct.pack('<d', 1.5) + bytes([0x10, 0x03]) + bytes([0x2A, len(summary)]) + summary
    assert tfe.encode_event(1.5, step=3, scalars=[('lr', 0.5)]) == want
    # first record of a file: wall_time + file_version (field 3)
    assert tfe.encode_event(2.0, file_version='brain.Event:2') == bytes([0x09]) + struct.pack('<d', 2.0) + bytes([0x1A, 13]) + b'brain.Event:2'
    # steps beyond 7 bits are multi-byte varints: 300 = 0xAC 0x02
    assert tfe.encode_event(0.0, step=300)[9:] == bytes([0x10, 0xAC, 0x02])


def test_record_framing():
    payload = b'abc'
    rec = tfe.frame(payload)
    assert rec[:8] == struct.pack('<Q', 3) and rec[12:15] == payload and len(rec) == 8 + 4 + 3 + 4
    assert struct.unpack('<I', rec[8:12])[0] == tfe.masked_crc(rec[:8])
    assert struct.unpack('<I', rec[15:])[0] == tfe.masked_crc(payload)


def test_writer_round_trip_with_the_reference_tags(tmp_path):
    w = tfe.SummaryWriter(str(tmp_path / 'log'))
    tags = ['lr', 'loss/loss_bce_1', 'loss/loss_dice_1', 'loss/loss_bce_2', 'loss/loss_dice_2', 'loss/loss_consistency', 'loss/loss_rec']
    for it in (0, 20, 400):
        for k, t in enumerate(tags):
            w.add_scalar(t, 0.25 * k + it, it)
    w.close()
    files = os.listdir(tmp_path / 'log')
    assert len(files) == 1 and files[0].startswith('events.out.tfevents.')
    ev = tfe.read_events(os.path.join(tmp_path / 'log', files[0]))
    assert ev[0]['file_version'] == 'brain.Event:2' and ev[0]['scalars'] == []
    got = [(e['step'], e['scalars'][0][0], e['scalars'][0][1]) for e in ev[1:]]
    assert got == [(it, t, 0.25 * k + it) for it in (0, 20, 400) for k, t in enumerate(tags)]
    assert all(e['wall_time'] > 1.6e9 for e in ev)


def test_reader_rejects_a_flipped_byte(tmp_path):
    w = tfe.SummaryWriter(str(tmp_path))
    w.add_scalar('lr', 1e-3, 7)
    w.close()
    raw = bytearray(open(w.path, 'rb').read())
    raw[-6] ^= 0x01
    open(w.path, 'wb').write(bytes(raw))
    with pytest.raises(ValueError):
        tfe.read_events(w.path)
