"""Real-data loader on the device: CIFAR-10 from the original python-pickle batches and from a TFRecord file
in the reference converter's layout (data/cifar_tfrecords.py:27-32: 3072 HWC bytes under key `image`), through the dataset plugin
into HBM-resident batches, and `train.py --dataset cifar` end to end on them."""
import os
import pickle
import struct
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import ROOT, pkg

pytestmark = pytest.mark.gpu


def write_pickles(root, rng, per_batch=40):
    d = os.path.join(root, 'cifar-10-batches-py')
    os.makedirs(d)
    imgs = []
    for i in range(1, 6):
        chw = rng.integers(0, 256, (per_batch, 3, 32, 32), dtype=np.uint8)          # the pickles hold planar CHW rows
        with open(os.path.join(d, 'data_batch_%d' % i), 'wb') as f:
            pickle.dump({b'data': chw.reshape(per_batch, 3072), b'labels': [0] * per_batch}, f)
        imgs.append(chw.transpose(0, 2, 3, 1))
    return np.concatenate(imgs)


def example_bytes(img):
    """tf.train.Example{features{feature{key 'image', bytes_list{value}}}} assembled from the protobuf wire format by hand."""
    def ld(field, payload):
        n, out = len(payload), bytearray([field << 3 | 2])
        while True:
            out.append((n & 0x7f) | (0x80 if n > 0x7f else 0))
            n >>= 7
            if not n:
                break
        return bytes(out) + payload
    feature = ld(1, ld(1, img.tobytes()))                    # Feature.bytes_list(1) -> BytesList.value(1)
    entry = ld(1, b'image') + ld(2, feature)                 # map entry: key(1), value(2)
    return ld(1, ld(1, entry))                               # Example.features(1) -> Features.feature(1)


def masked_crc32c(data):
    poly, crc = 0x82F63B78, 0xFFFFFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ (poly if crc & 1 else 0)
    crc ^= 0xFFFFFFFF
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


@pytest.mark.parametrize('form', ['pickle', 'tfrecord'])
def test_cifar_batches_reach_the_device_unchanged(tmp_path, form):
    rt, K, datasets = pkg('runtime'), pkg('kernels'), pkg('datasets')
    rng = np.random.default_rng(9)
    root = str(tmp_path)
    if form == 'pickle':
        imgs = write_pickles(root, rng)
    else:
        imgs = rng.integers(0, 256, (64, 32, 32, 3), dtype=np.uint8)
        with open(os.path.join(root, 'cifar.32.train.tfrecords'), 'wb') as f:      # framing: u64 length, masked CRC-32C of it, payload, its CRC
            for im in imgs:
                p = example_bytes(im)
                head = struct.pack('<Q', len(p))
                f.write(head + struct.pack('<I', masked_crc32c(head)) + p + struct.pack('<I', masked_crc32c(p)))
    sess = rt.Session(device=torch.device('cuda:0'), dtype=K.BF16, seed=0, rank=0, world_size=1)
    args = SimpleNamespace(dataset='cifar', dataset_dir=root, data_dir=root, batch_size=16, shuffle=False, seed=0, resize=None, grayscale=False)
    src, n, shape = datasets.get_dataset(args, sess)
    assert (n, shape) == (len(imgs), (32, 32, 3))
    for i in range(len(imgs) // 16):
        b = src.next_batch()
        assert b.is_cuda and tuple(b.shape) == (16, 32, 32, 3) and b.dtype == torch.float32
        assert np.abs(b.cpu().numpy() - imgs[16 * i:16 * (i + 1)].astype(np.float32) / 255.0).max() < 1e-7     # (one ulp: the device divides by 255 its own way)


def test_train_cli_on_cifar_pickles(tmp_path):
    """train.py on the pickles: iter_per_epoch = N // (B * n_gpus) (train.py:222) and finite losses."""
    root = str(tmp_path)
    write_pickles(root, np.random.default_rng(1))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), '--model', 'iwgan', '--batch_size', '8', '--latent_size', '16',
                        '--optimizer', 'adam', '--lr', '1e-4', '--beta1', '0.5', '--beta2', '0.9', '--dataset', 'cifar', '--dataset_dir', root,
                        '--n_disc_train', '1', '--epochs', '1', '--dir', os.path.join(root, 'ws')],
                       env=env, timeout=900, capture_output=True, text=True)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    out = p.stdout + p.stderr
    assert '25/25' in out.replace(' ', ''), out[-1500:]          # 200 images // 8
    assert 'nan' not in out.lower().split('starting training')[-1]


def test_train_cli_pix2pix_on_nyuv2_png_records(tmp_path):
    """Config 4's real input path end to end (hem/data/nyuv2.py:148-262 -> hem/models/pix2pix.py): PNG records (8-bit RGB image,
    16-bit depth) in `nyuv2.train.tfrecords`, decoded by the plugin, one 256 x 256 crop window per pair, served from HBM to the
    pix2pix step through `train.py @config`-style flags."""
    from test_host_gen2 import _png_encode
    tfr = pkg('tfrecord')
    rng = np.random.default_rng(4)
    recs = []
    for i in range(5):
        rgb = rng.integers(0, 256, (260, 264, 3), dtype=np.uint8)
        depth = rng.integers(1000, 60000, (260, 264, 1), dtype=np.uint16)
        recs.append(tfr.make_example({'image': _png_encode(rgb, [1, 2, 4]), 'depth': _png_encode(depth, [2, 0]),
                                      'width': 260, 'height': 264, 'channels': 3}))
    d = tmp_path / 'datasets'
    d.mkdir()
    tfr.write_records(str(d / 'nyuv2.train.tfrecords'), recs)
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), '--model', 'pix2pix', '--dataset', 'nyuv2', '--dataset_dir', str(d),
                        '--cache_dir', str(tmp_path / 'cache'), '--random_crop', '256', '256', '--batch_size', '2', '--optimizer', 'adam',
                        '--lr', '1e-4', '--beta1', '0.5', '--n_disc_train', '1', '--skip_layers', '--epochs', '1',
                        '--dir', str(tmp_path / 'ws')], env=env, timeout=900, capture_output=True, text=True)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    out = p.stdout + p.stderr
    assert 'Training complete' in out and '2/2' in out.replace(' ', ''), out[-1500:]          # 5 pairs // 2
    assert 'nan' not in out.lower().split('starting training')[-1]


def test_streaming_source_over_the_pinned_ring_reaches_the_device_unchanged():
    """data.StreamingSource on the GPU box: shuffle-buffer draws on the host, batches assembled in pinned memory, asynchronous
    copies on a side stream, float32 in [0, 1] on the device -- every batch equals the host array's rows at the drawn indices,
    across many reuses of the three ring slots, with kernels in flight on the compute stream meanwhile."""
    data = pkg('data')
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(2)
    imgs = rng.integers(0, 256, (5000, 32, 32, 3), dtype=np.uint8)
    src = data.StreamingSource(imgs, 512, dev, buffer_size=1000, seed=11)
    busy = torch.randn(2048, 2048, device=dev)
    for step in range(40):
        busy = (busy @ busy).tanh()                                            # work on the compute stream while copies land
        x, idx = src.next_batch(return_indices=True)
        assert x.is_cuda and x.dtype == torch.float32 and tuple(x.shape) == (512, 32, 32, 3)
        assert np.abs(x.cpu().numpy() - imgs[idx].astype(np.float32) / 255.0).max() < 1e-7
    src.close()


def test_train_cli_streaming_pipeline(tmp_path):
    """`train.py --dataset cifar --streaming`: the dataset stays in host memory and reaches the replica through the shuffle buffer
    and the pinned ring (what a dataset beyond --hbm_budget_gb takes by itself)."""
    root = str(tmp_path)
    write_pickles(root, np.random.default_rng(1))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'train.py'), '--model', 'iwgan', '--batch_size', '8', '--latent_size', '16',
                        '--optimizer', 'adam', '--lr', '1e-4', '--beta1', '0.5', '--beta2', '0.9', '--dataset', 'cifar', '--dataset_dir', root,
                        '--streaming', '--buffer_size', '50', '--n_disc_train', '1', '--epochs', '1', '--dir', os.path.join(root, 'ws')],
                       env=env, timeout=900, capture_output=True, text=True)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    out = p.stdout + p.stderr
    assert '25/25' in out.replace(' ', ''), out[-1500:]
    assert 'nan' not in out.lower().split('starting training')[-1]
