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
