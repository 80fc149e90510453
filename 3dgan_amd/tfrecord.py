"""Minimal TFRecord + tf.train.Example reader/writer (no TensorFlow): the on-disk format the
reference's dataset scripts write (data/cifar_tfrecords.py:19-36, hem/data/mnist.py:66-77) and
its input pipeline reads (data.py:34-60).  Host I/O next to the hot path.

Record framing: uint64 length | uint32 masked-crc32c(length) | payload | uint32 masked-crc32c(payload).
Example payload: protobuf `Features{ map<string, Feature> feature = 1 }`, Feature.bytes_list = 1,
Feature.int64_list = 3.  Only what those files contain is parsed.
"""
import struct

import numpy as np

_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t.append(c)
        _CRC_TABLE = t
    return _CRC_TABLE


def crc32c(data):
    t = _crc_table()
    c = 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def read_records(path, verify=False):
    """Yield the raw payload of every record."""
    with open(path, 'rb') as f:
        while True:
            head = f.read(12)
            if len(head) < 12:
                return
            n, = struct.unpack('<Q', head[:8])
            if verify and struct.unpack('<I', head[8:])[0] != masked_crc(head[:8]):
                raise IOError('%s: corrupt record header' % path)
            payload = f.read(n)
            tail = f.read(4)
            if len(payload) < n or len(tail) < 4:
                raise IOError('%s: truncated record' % path)
            if verify and struct.unpack('<I', tail)[0] != masked_crc(payload):
                raise IOError('%s: corrupt record payload' % path)
            yield payload


def _varint(buf, i):
    v, s = 0, 0
    while True:
        b = buf[i]
        i += 1
        v |= (b & 0x7F) << s
        if not b & 0x80:
            return v, i
        s += 7


def _fields(buf):
    """Iterate (field_number, wire_type, value) over one protobuf message."""
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 2:
            ln, i = _varint(buf, i)
            v = buf[i:i + ln]
            i += ln
        elif wt == 1:
            v = buf[i:i + 8]
            i += 8
        elif wt == 5:
            v = buf[i:i + 4]
            i += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        yield fn, wt, v


def parse_example(payload):
    """{feature name: bytes | list of ints} of a tf.train.Example."""
    out = {}
    for fn, _, features in _fields(payload):
        if fn != 1:
            continue
        for fn2, _, entry in _fields(features):               # map entries
            if fn2 != 1:
                continue
            key, feat = None, None
            for fn3, _, v in _fields(entry):
                if fn3 == 1:
                    key = bytes(v).decode()
                elif fn3 == 2:
                    feat = v
            if key is None or feat is None:
                continue
            for kind, _, lst in _fields(feat):
                if kind == 1:                                # bytes_list
                    vals = [bytes(v) for f4, _, v in _fields(lst) if f4 == 1]
                    out[key] = vals[0] if len(vals) == 1 else vals
                elif kind == 3:                              # int64_list (packed or not)
                    ints = []
                    for f4, wt, v in _fields(lst):
                        if f4 != 1:
                            continue
                        if wt == 0:
                            ints.append(v)
                        else:
                            j = 0
                            while j < len(v):
                                x, j = _varint(v, j)
                                ints.append(x)
                    out[key] = ints
    return out


# ---- writer (tests and dataset conversion) ---------------------------------------------------------
def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(fn, payload):
    return _enc_varint((fn << 3) | 2) + _enc_varint(len(payload)) + payload


def make_example(features):
    """features: {name: bytes | int}."""
    entries = b''
    for k, v in features.items():
        if isinstance(v, (bytes, bytearray)):
            feat = _ld(1, _ld(1, bytes(v)))
        else:
            feat = _ld(3, _enc_varint((1 << 3) | 0) + _enc_varint(int(v)))
        entries += _ld(1, _ld(1, k.encode()) + _ld(2, feat))
    return _ld(1, entries)


def write_records(path, payloads):
    with open(path, 'wb') as f:
        for p in payloads:
            head = struct.pack('<Q', len(p))
            f.write(head + struct.pack('<I', masked_crc(head)) + p + struct.pack('<I', masked_crc(p)))


def load_image_tfrecords(path, shape, key='image'):
    """All records' raw uint8 image bytes as one array [N, *shape] (CIFAR: 3072 HWC bytes)."""
    n = int(np.prod(shape))
    imgs = [np.frombuffer(parse_example(p)[key], dtype=np.uint8) for p in read_records(path)]
    for im in imgs:
        if im.size != n:
            raise ValueError('%s: record holds %d bytes, expected %d' % (path, im.size, n))
    return np.stack(imgs).reshape((-1,) + tuple(shape))
