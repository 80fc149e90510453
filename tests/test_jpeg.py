"""The native JPEG decoder (tdg_jpeg_decode, 3dgan_amd/jpeg.py) behind the floorplan plugin: the reference decodes the
records' raw file bytes with `tf.image.decode_image(..., channels=3)` (data.py:15; data/floorplan_tfrecords.py:26-41 writes
them).  TensorFlow links libjpeg; Pillow links the same decoder family with the same defaults (accurate integer inverse
DCT, triangle chroma upsampling), so Pillow's decode of files Pillow wrote is the yardstick here: BIT-EXACT.  Host work
only -- no GPU."""
import importlib
import io
import os
import struct

import numpy as np
import pytest

PIL_Image = pytest.importorskip('PIL.Image')

jpeg = importlib.import_module('3dgan_amd.jpeg')
_lib = importlib.import_module('3dgan_amd._lib')


def _picture(h, w, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.stack([127 + 100 * np.sin(xx / 7.0 + yy / 11.0), 127 + 90 * np.cos(xx / 5.0), 127 + 80 * np.sin(yy / 3.0 + 1)], -1)
    a += rng.normal(0, 12, a.shape)
    return np.clip(a, 0, 255).astype(np.uint8)


def _encode(arr, **kw):
    b = io.BytesIO()
    PIL_Image.fromarray(arr).save(b, 'JPEG', **kw)
    return b.getvalue()


def _pil(data):
    return np.asarray(PIL_Image.open(io.BytesIO(data)).convert('RGB'))


@pytest.mark.parametrize('hw', [(64, 64), (37, 53), (8, 8), (1, 1), (17, 300), (129, 65), (2, 2)])
@pytest.mark.parametrize('subsampling', [0, 1, 2])                       # 4:4:4, 4:2:2, 4:2:0
def test_colour_files_decode_bit_exactly(hw, subsampling):
    for q in (35, 75, 95):
        data = _encode(_picture(*hw, seed=q), quality=q, subsampling=subsampling)
        assert jpeg.is_jpeg(data)
        got = jpeg.decode(data)
        assert got.shape == (hw[0], hw[1], 3) and got.dtype == np.uint8
        assert np.array_equal(got, _pil(data)), (hw, subsampling, q)


def test_grayscale_is_replicated_to_three_channels():
    data = _encode(_picture(40, 52)[..., 0], quality=80)
    got = jpeg.decode(data)
    assert np.array_equal(got, _pil(data)) and np.array_equal(got[..., 0], got[..., 2])


def test_optimised_huffman_tables_and_restart_intervals():
    pic = _picture(70, 90, seed=3)
    data = _encode(pic, quality=85, optimize=True)
    assert np.array_equal(jpeg.decode(data), _pil(data))
    try:
        data = _encode(pic, quality=85, subsampling=2, restart_marker_blocks=3)
    except TypeError:
        pytest.skip('this Pillow cannot write restart markers')
    if b'\xff\xdd' not in data:
        pytest.skip('this Pillow ignored restart_marker_blocks')
    assert np.array_equal(jpeg.decode(data), _pil(data))


def test_unsupported_files_are_refused_by_name():
    data = _encode(_picture(32, 32), quality=80, progressive=True)
    with pytest.raises(_lib.TdgError, match='progressive'):
        jpeg.decode(data)
    with pytest.raises(_lib.TdgError, match='SOI'):
        jpeg.decode(b'\x89PNG\r\n\x1a\n' + b'\0' * 32)
    ok = _encode(_picture(32, 32), quality=80)
    with pytest.raises(_lib.TdgError):
        jpeg.decode(ok[:len(ok) // 8])                              # cut inside the headers


def _segment(data, marker):
    """(offset of the segment's first payload byte, payload length) of the first `marker` segment."""
    i = 2
    while i + 4 <= len(data):
        assert data[i] == 0xff
        m, ln = data[i + 1], (data[i + 2] << 8) | data[i + 3]
        if m == marker:
            return i + 4, ln - 2
        i += 2 + ln
    raise AssertionError('no segment %02x' % marker)


def test_corrupt_headers_are_refused_not_crashed_on():
    """an over-subscribed Huffman table (bits[1] = 255 passes the count check) used to write far past the
    9-bit lookahead table.  Mutated DHT / SOF / SOS bytes must come back as errors (or decode), never as a crash."""
    data = bytearray(_encode(_picture(24, 24), quality=80))
    off, _ = _segment(data, 0xc4)
    n = sum(data[off + 1:off + 17])                           # values of the first table: keep the count, so the segment parses
    assert 4 <= n <= 255
    bad = bytearray(data)
    bad[off + 1:off + 17] = bytes([n] + [0] * 15)             # n codes of length 1 (there are two)
    with pytest.raises(_lib.TdgError, match='bogus Huffman table'):
        jpeg.decode(bytes(bad))
    bad = bytearray(data)
    bad[off + 1:off + 17] = bytes([1, 3, n - 4] + [0] * 13)   # 1 code of length 1, then 3 of length 2 (only 2 are left)
    with pytest.raises(_lib.TdgError, match='bogus Huffman table'):
        jpeg.decode(bytes(bad))
    # the advisor's file: a table of 255 one-bit codes with its 255 values present
    big = bytes([0xff, 0xd8, 0xff, 0xc4]) + struct.pack('>H', 2 + 17 + 255) + bytes([0x00, 255] + [0] * 15) + bytes(range(255)) + bytes(data[2:])
    with pytest.raises(_lib.TdgError, match='bogus Huffman table'):
        jpeg.decode(big)
    rng = np.random.default_rng(5)
    for marker in (0xc4, 0xc0, 0xda, 0xdb):
        off, ln = _segment(data, marker)
        for _ in range(150):
            bad = bytearray(data)
            for _ in range(int(rng.integers(1, 4))):
                bad[off + int(rng.integers(0, ln))] = int(rng.integers(0, 256))
            try:
                out = jpeg.decode(bytes(bad))
                assert out.ndim == 3 and out.shape[2] == 3
            except _lib.TdgError:
                pass


def test_grayscale_with_sampling_factors_in_the_frame_header():
    """a one-component file whose SOF says 2x2 is still one block per MCU (non-interleaved scan);
    libjpeg (Pillow) decodes it exactly like the 1x1 file."""
    data = bytearray(_encode(_picture(40, 52)[..., 0], quality=80))
    off, _ = _segment(data, 0xc0)
    assert data[off + 5] == 1 and data[off + 7] == 0x11
    data[off + 7] = 0x22
    assert np.array_equal(jpeg.decode(bytes(data)), _pil(bytes(data)))


def test_truncated_scan_is_an_error():
    """tf.image.decode_image raises on a file cut inside its entropy-coded data; no grey tail with a success status."""
    data = _encode(_picture(64, 64), quality=90)
    with pytest.raises(_lib.TdgError, match='before the last block'):
        jpeg.decode(data[:int(len(data) * 0.6)])
    assert np.array_equal(jpeg.decode(data), _pil(data))          # the whole file still decodes
    assert np.array_equal(jpeg.decode(data[:-2]), _pil(data))     # ... also without its EOI marker (every block is there)


def test_floorplan_plugin_reads_jpeg_records(tmp_path):
    """A floorplans.train.tfrecords file framed here byte by byte (tf.train.Example: `image` = the JPEG file's bytes) ->
    the plugin's 64 x 64 uint8 images == the same pipeline on Pillow's decode."""
    tfrecord = importlib.import_module('3dgan_amd.tfrecord')
    fp = importlib.import_module('3dgan_amd.data_plugins.floorplan')
    common = importlib.import_module('3dgan_amd.data_plugins._common')
    import torch
    from types import SimpleNamespace

    def varint(n):
        out = b''
        while True:
            b7 = n & 0x7f
            n >>= 7
            out += bytes([b7 | (0x80 if n else 0)])
            if not n:
                return out

    def bytes_feature(key, value):
        bl = b'\x0a' + varint(len(value)) + value                              # BytesList.value
        feat = b'\x0a' + varint(len(bl)) + bl                                  # Feature.bytes_list
        entry = b'\x0a' + varint(len(key)) + key + b'\x12' + varint(len(feat)) + feat
        return b'\x0a' + varint(len(entry)) + entry                            # Features.feature map entry

    files = [_encode(_picture(90, 120, seed=s), quality=90, subsampling=2) for s in range(3)]
    path = tmp_path / 'floorplans.train.tfrecords'
    with open(path, 'wb') as f:
        for data in files:
            feats = bytes_feature(b'image', data)
            ex = b'\x0a' + varint(len(feats)) + feats                          # Example.features
            f.write(struct.pack('<Q', len(ex)) + struct.pack('<I', tfrecord.masked_crc(struct.pack('<Q', len(ex)))))
            f.write(ex + struct.pack('<I', tfrecord.masked_crc(ex)))
    args = SimpleNamespace(dataset_dir=str(tmp_path), cache_dir=None)
    imgs = fp.FloorplanDataset.load(args)
    assert imgs.shape == (3, 64, 64, 3) and imgs.dtype == np.uint8
    for got, data in zip(imgs, files):
        x = torch.from_numpy(_pil(data).astype(np.float32))[None]
        want = np.clip(np.rint(common.resize_bilinear_tf1(x, 64, 64)[0].numpy()), 0, 255).astype(np.uint8)
        assert np.array_equal(got, want)
