"""Host side of the gen-2 boundary: `@file` configs, the three-pass parse with plugin flags,
plugin discovery by base-class name, and the input formats either side of the hot path (TFRecord framing, PNG records,
the nyuv2 / floorplan parsers).  No GPU."""
import importlib
import os
import struct
import sys
import zlib
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import pkg, ROOT

# the data of the reference's examples/pix2pix.config and examples/pix2pix/noise.config (key / value lines)
PIX2PIX_CONFIG = """model\t\t pix2pix
epochs\t\t 50
batch_size \t 64
examples     64
n_gpus \t\t 2
optimizer\t adam
lr\t\t\t 1e-4
beta1\t\t 0.5
# beta2        0.9
dataset\t\t nyuv2
cache_dir\t tmp/256x256
random_crop  256 256
n_disc_train 1
skip_layers
check_numerics
"""


def test_reference_pix2pix_config_parses(tmp_path):
    """`python train.py @examples/pix2pix.config` (hem/util/arguments.py:10-179): `@file` with comments, general flags,
    the nyuv2 plugin's --random_crop (pass 2), the pix2pix plugin's flags (pass 3)."""
    A = pkg('arguments')
    cfg = tmp_path / 'pix2pix.config'
    cfg.write_text(PIX2PIX_CONFIG)
    warned = []
    a = A.parse_args(['@' + str(cfg), '--dir', 'w'], warn=warned.append)
    assert warned == [] and a.unknown_args == []
    assert (a.model, a.epochs, a.batch_size, a.examples, a.n_gpus, a.optimizer, a.lr, a.beta1, a.beta2) == \
        ('pix2pix', '50', 64, 64, 2, 'adam', 1e-4, 0.5, 0.999)
    assert (a.dataset, a.cache_dir, a.random_crop, a.n_disc_train, a.check_numerics, a.dir) == \
        ('nyuv2', 'tmp/256x256', [256, 256], 1, True, 'w')
    assert a.skip_layers is True and a.noise == [] and a.dropout == 0 and a.add_l1 is False and getattr(a, 'lambda') == 10.0
    # gen-2 general flags and their defaults (hem/util/arguments.py:78-150)
    assert (a.max_to_keep, a.test_epochs, a.raw_dataset_dir, a.dataset_dir) == (0, [], '/tmp', 'datasets') and a.n_threads >= 1
    # the command line wins over the file; plugin flags are known on the command line too
    b = A.parse_args(['@' + str(cfg), '--dataset', 'synthetic', '--lr', '0.1', '--noise', 'input', 'end', '--dropout', '0.5'],
                     warn=warned.append)
    assert (b.dataset, b.lr, b.noise, b.dropout) == ('synthetic', 0.1, ['input', 'end'], 0.5) and warned == []


def test_unknown_arguments_warn_like_the_reference(tmp_path):
    """hem/util/arguments.py:161-163: leftovers are a WARNING.  examples/pix2pix/noise.config carries the retired flag
    `add_noise1`, so the reference's own example depends on this."""
    A = pkg('arguments')
    cfg = tmp_path / 'noise.config'
    cfg.write_text(PIX2PIX_CONFIG + 'dropout      0\nadd_noise1\n')
    warned = []
    a = A.parse_args(['@' + str(cfg)], warn=warned.append)
    assert a.unknown_args == ['--add_noise1'] and len(warned) == 1 and 'add_noise1' in warned[0]
    assert a.dropout == 0.0
    with pytest.raises(SystemExit):
        A.parse_args(['--model', 'no_such_model'], warn=warned.append)
    # gen-1 models take no plugin pass; their n_disc_train default is train.py:107-111's 5
    g = A.parse_args(['--model', 'iwgan', '--data', 'CIFAR'], warn=warned.append)
    assert (g.model, g.dataset, g.n_disc_train) == ('iwgan', 'cifar', 5)
    assert (g.wgan_clip, g.gp_per_sample, g.vae_full_elbo, g.mean_loss) == (0.0, False, False, False)


def test_plugin_discovery_by_first_base_name(tmp_path):
    """hem/util/data.py:11-29: classes DEFINED in a module of the plugin directory whose FIRST base is named
    `ModelPlugin` are registered under `name`; helpers, imported classes and second bases are not."""
    P = pkg('plugins')
    assert set(P.model_plugins()) == {'pix2pix'}
    assert P.get_model('pix2pix').__name__ == 'pix2pix'
    assert {'cifar', 'mnist', 'floorplan', 'nyuv2', 'synthetic'} <= set(P.data_plugins())
    assert pkg('models').get_model('pix2pix') is P.get_model('pix2pix')
    with pytest.raises(KeyError):
        P.get_model('nope')
    d = tmp_path / 'plug_pkg'
    d.mkdir()
    (d / '__init__.py').write_text('')
    (d / 'ModelPlugin.py').write_text('class ModelPlugin:\n    name = None\n')
    (d / 'mine.py').write_text(
        'from .ModelPlugin import ModelPlugin\n'
        'class Helper:\n    name = "helper"\n'
        'class Mine(ModelPlugin):\n    name = "mine"\n'
        'class Second(Helper, ModelPlugin):\n    name = "second"\n')
    sys.path.insert(0, str(tmp_path))
    try:
        found = P.search_for_plugins(str(d), 'plug_pkg', 'ModelPlugin')
    finally:
        sys.path.remove(str(tmp_path))
    assert set(found) == {'mine'} and found['mine'].__name__ == 'Mine'


# ------------------------------------------------------------------------------------------------ byte formats
def _crc32c_bitwise(data):
    """CRC-32C (Castagnoli), reflected polynomial 0x82F63B78, one bit at a time: an independent statement."""
    crc = 0xFFFFFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 & -(crc & 1))
    return crc ^ 0xFFFFFFFF


def _masked(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def test_tfrecord_reader_on_a_hand_assembled_file(tmp_path):
    """A TFRecord file built byte by byte from the documented framing (u64 length, masked CRC-32C of the length, payload,
    masked CRC-32C of the payload) around a hand-encoded tf.train.Example -- NOT by this repo's writer."""
    tfr = pkg('tfrecord')
    img = bytes(range(256)) * 12                                              # 3072 bytes: one CIFAR image
    # Example{ features{ feature{ key:"image" value{ bytes_list{ value: img } } } feature{ key:"label" value{ int64_list{ value: 7 } } } } }
    bytes_list = b'\x0a' + b'\x80\x18' + img                                  # field 1, len 3072 = 0x80 0x18 (varint)
    feat_img = b'\x0a' + b'\x83\x18' + bytes_list                             # Feature.bytes_list, len 3075
    entry_img = b'\x0a\x05image' + b'\x12' + b'\x86\x18' + feat_img           # map entry: key, value (len 3078)
    int_list = b'\x08\x07'                                                    # Int64List.value (unpacked varint 7)
    feat_lab = b'\x1a\x02' + int_list
    entry_lab = b'\x0a\x05label' + b'\x12\x04' + feat_lab
    features = b'\x0a' + b'\x90\x18' + entry_img + b'\x0a\x0d' + entry_lab    # two map entries (3088 and 13 bytes)
    assert len(entry_img) == 3088 and len(entry_lab) == 13
    example = b'\x0a' + b'\xa2\x18' + features                                # Example.features, len 3106
    assert len(features) == 3106
    head = struct.pack('<Q', len(example))
    blob = head + struct.pack('<I', _masked(_crc32c_bitwise(head))) + example + struct.pack('<I', _masked(_crc32c_bitwise(example)))
    path = tmp_path / 'hand.tfrecords'
    path.write_bytes(blob * 2)
    recs = list(tfr.read_records(str(path), verify=True))
    assert len(recs) == 2
    ex = tfr.parse_example(recs[1])
    assert ex['image'] == img and ex['label'] == [7]
    assert np.array_equal(tfr.load_image_tfrecords(str(path), (32, 32, 3))[0].reshape(-1), np.frombuffer(img, np.uint8))
    bad = bytearray(blob)
    bad[20] ^= 1                                                              # a flipped payload bit fails the CRC
    (tmp_path / 'bad.tfrecords').write_bytes(bytes(bad))
    with pytest.raises(IOError, match='corrupt record payload'):
        list(tfr.read_records(str(tmp_path / 'bad.tfrecords'), verify=True))


def _png_encode(img, filters):
    """Test-side PNG encoder (8/16-bit grey or RGB) that applies the given filter type per scanline (cycled)."""
    img = np.asarray(img)
    h, w, c = img.shape
    depth = 16 if img.dtype == np.uint16 else 8
    raw = img.astype('>u2').tobytes() if depth == 16 else img.tobytes()
    bpp = c * depth // 8
    rb = w * bpp
    rows = [np.frombuffer(raw[r * rb:(r + 1) * rb], np.uint8).astype(np.int32) for r in range(h)]
    out = bytearray()
    for r in range(h):
        ft = filters[r % len(filters)]
        cur, up = rows[r], rows[r - 1] if r else np.zeros(rb, np.int32)
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        ul = np.concatenate([np.zeros(bpp, np.int32), up[:-bpp]])
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - left
        elif ft == 2:
            f = cur - up
        elif ft == 3:
            f = cur - ((left + up) >> 1)
        else:
            p = left + up - ul
            pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
            f = cur - pred
        out.append(ft)
        out += (f & 0xFF).astype(np.uint8).tobytes()

    def chunk(tag, data):
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)
    color = {1: 0, 3: 2}[c]
    return (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, depth, color, 0, 0, 0)) +
            chunk(b'IDAT', zlib.compress(bytes(out), 6)[:100]) + chunk(b'IDAT', zlib.compress(bytes(out), 6)[100:]) +
            chunk(b'IEND', b''))


def test_png_decoder_all_filter_types_8_and_16_bit():
    """The native scanline reconstruction (tdg_png_unfilter) on every PNG filter type, split IDAT chunks, 8-bit RGB and
    16-bit grey (the two formats of the nyuv2 records, hem/data/nyuv2.py:128-153)."""
    png = pkg('png')
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (13, 17, 3), dtype=np.uint8)
    rgb[4:9, 3:12] = 200                                                      # flat area: filters produce zeros
    d16 = rng.integers(0, 65536, (11, 9, 1), dtype=np.uint16)
    for filters in ([0], [1], [2], [3], [4], [4, 3, 2, 1, 0]):
        assert np.array_equal(png.decode(_png_encode(rgb, filters)), rgb), filters
        out = png.decode(_png_encode(d16, filters))
        assert out.dtype == np.uint16 and np.array_equal(out, d16), filters
    assert png.decode(_png_encode(d16, [4]), channels=3).shape == (11, 9, 3)
    assert png.decode(_png_encode(rgb, [1]), channels=1).shape == (13, 17, 1)
    assert np.array_equal(png.decode(pkg('summaries').png_bytes(rgb / 255.0)), rgb)       # this repo's own writer too
    with pytest.raises(ValueError, match='jpeg'):
        png.decode(b'\xff\xd8\xff\xe0' + b'\0' * 32)
    corrupt = bytearray(_png_encode(rgb, [0]))
    corrupt[40] ^= 0xFF
    with pytest.raises(ValueError):
        png.decode(bytes(corrupt))


def test_resize_bilinear_tf1_formula():
    """tf.image.resize_images of TF 1.x: src = dst * in / out (no half-pixel centres), bilinear, edge clamped."""
    C = pkg('data_plugins._common')
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 255, (2, 7, 5, 3))
    out = C.resize_bilinear_tf1(torch.tensor(x), 4, 9).numpy()
    ref = np.zeros((2, 4, 9, 3))
    for oy in range(4):
        sy = oy * 7 / 4
        y0 = int(np.floor(sy)); y1 = min(y0 + 1, 6); fy = sy - y0
        for ox in range(9):
            sx = ox * 5 / 9
            x0 = int(np.floor(sx)); x1 = min(x0 + 1, 4); fx = sx - x0
            top = x[:, y0, x0] * (1 - fx) + x[:, y0, x1] * fx
            bot = x[:, y1, x0] * (1 - fx) + x[:, y1, x1] * fx
            ref[:, oy, ox] = top * (1 - fy) + bot * fy
    assert np.allclose(out, ref, atol=1e-9)


def test_nyuv2_plugin_reads_png_records_crops_and_drops_sensor_gaps(tmp_path):
    """hem/data/nyuv2.py:148-262 end to end on a small hand-made `nyuv2.train.tfrecords`: decode both PNGs, one crop
    window per pair (same for image and depth), image / 255, depth / 65535, pairs whose depth crop holds 0 or 65535 are
    never served."""
    tfr, P = pkg('tfrecord'), pkg('plugins')
    rng = np.random.default_rng(2)
    H, W, n = 24, 32, 6
    recs, rgbs, depths = [], [], []
    for i in range(n):
        rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        depth = rng.integers(1000, 60000, (H, W, 1), dtype=np.uint16)
        if i == 2:
            depth[:, :] = 0                                                   # a frame the sensor never returned
        if i == 4:
            depth[5, 7] = 65535                                               # one saturated pixel
        rgbs.append(rgb)
        depths.append(depth)
        recs.append(tfr.make_example({'image': _png_encode(rgb, [4, 1]), 'depth': _png_encode(depth, [2, 3]),
                                      'width': H, 'height': W, 'channels': 3}))
    d = tmp_path / 'datasets'
    d.mkdir()
    tfr.write_records(str(d / 'nyuv2.train.tfrecords'), recs)
    args = SimpleNamespace(dataset_dir=str(d), data_dir='data', cache_dir=str(tmp_path / 'cache'), batch_size=4, seed=3,
                           random_crop=[16, 16], resize=None, include_location=False, normalize=False, include_originals=None)
    sess = SimpleNamespace(device=torch.device('cpu'), rank=0, world_size=1)
    plug = P.get_dataset('nyuv2')
    assert plug.check_prepared_datasets(str(d)) is False                      # validate / test files are missing
    src, count, shape = plug.get_source(args, sess)
    assert (count, shape) == (n, (16, 16, 3))
    assert os.path.exists(tmp_path / 'cache' / 'nyuv2.train.npz')
    rgb_all, dep_all = np.stack(rgbs), np.stack(depths)[..., 0]
    seen = set()
    for _ in range(6):
        x, y = src.next_batch()
        assert x.shape == (4, 16, 16, 3) and y.shape == (4, 16, 16, 1) and x.dtype == torch.float32
        assert float(y.min()) > 0.0 and float(y.max()) < 1.0
        for b in range(4):
            # locate the crop: the window of some frame that matches exactly, depth at the SAME window
            xb = np.rint(x[b].numpy() * 255).astype(np.uint8)
            hit = None
            for i in range(n):
                for t in range(H - 15):
                    for l in range(W - 15):
                        if np.array_equal(rgb_all[i, t:t + 16, l:l + 16], xb):
                            hit = (i, t, l)
            assert hit is not None
            i, t, l = hit
            seen.add(i)
            assert i != 2 and not (i == 4 and t <= 5 < t + 16 and l <= 7 < l + 16)
            assert np.allclose(y[b, :, :, 0].numpy(), dep_all[i, t:t + 16, l:l + 16] / 65535.0, atol=1e-7)
    assert seen >= {0, 1, 3, 5}
    # gen-1 entry point and alias
    ds = pkg('datasets')
    args.dataset, args.model = 'nyuv2', 'pix2pix'
    assert ds.get_dataset(args, sess)[1] == n


def test_residual_and_norm_builders():
    """hem/ops/layers.py:215-320 builder surface: variable names, batch-norm numbering, rejected forms."""
    Lm, act = pkg('ops.layers'), pkg('ops.activations')
    Lm.reset_graph()
    x = Lm.placeholder((None, 8, 8, 3))
    with Lm.variable_scope('generator') as net:
        h = Lm.conv2d(x, 3, 4, use_batch_norm=True, stride=1, name='a')
        h = Lm.residual(h, 4, 6, use_batch_renorm=True, activation=act.relu, name='r')
        h = Lm.conv2d(h, 6, 2, use_instance_norm=True, stride=2, name='b')
        h = Lm.deconv2d(h, 2, 2, use_batch_renorm=True, name='c')
    assert h.shape == (None, 8, 8, 2)
    kinds = [(l.kind, l.use_bn, l.use_in, l.n_bn) for l in net.layers]
    assert kinds == [('conv2d', True, False, 1), ('residual', True, False, 2), ('conv2d', False, True, 0), ('deconv2d', True, False, 1)]
    assert [net.bn_name(0, 0), net.bn_name(0, 1, 0), net.bn_name(0, 1, 1), net.bn_name(0, 3)] == \
        ['generator/BatchNorm/beta', 'generator/BatchNorm_1/beta', 'generator/BatchNorm_2/beta', 'generator/BatchNorm_3/beta']
    assert net.bn_name(1, 0) == 'generator/BatchNorm_4/beta'           # a second (reuse) pass continues the numbering
    with pytest.raises(ValueError):
        Lm.residual(h, 2, 2, stride=2, name='s')
    with pytest.raises(NotImplementedError):
        Lm.residual(h, 2, 2, use_instance_norm=True, name='t')
    with pytest.raises(NotImplementedError):
        Lm.conv2d(h, 2, 2, use_instance_norm=True, use_batch_norm=True, name='u')
