"""Input pipeline entry point with the contract of the reference's `data.py:34-60`
(`get_dataset(args) -> x, x_init, x_count`): here `(source, count, image_shape)` where `source`
serves per-replica batches already resident in HBM.

The reference's CIFAR parsers are broken, so the loader reads the bytes the
dataset scripts actually wrote: 3072 HWC uint8 per record -> float32 / 255 (data/cifar_tfrecords.py:27-32).
`--resize W H` and `--grayscale` (train.py:226-231) are applied once on the device.
"""
import gzip
import os
import pickle
import struct

import numpy as np
import torch
import torch.nn.functional as F

from . import tfrecord
from .data import ArraySource, SyntheticSource, SyntheticPairSource


def _cifar(data_dir):
    tfr = os.path.join(data_dir, 'cifar.32.train.tfrecords')                 # data.py:39
    if os.path.exists(tfr):
        return tfrecord.load_image_tfrecords(tfr, (32, 32, 3))
    pk = os.path.join(data_dir, 'cifar-10-batches-py')
    if os.path.isdir(pk):                                                    # data/cifar_tfrecords.py:23-29
        out = []
        for i in range(1, 6):
            with open(os.path.join(pk, 'data_batch_%d' % i), 'rb') as f:
                d = pickle.load(f, encoding='bytes')
            out.append(d[b'data'].reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1))
        return np.concatenate(out)
    raise FileNotFoundError('no CIFAR-10 data under %s (expected cifar.32.train.tfrecords or cifar-10-batches-py/); '
                            'use --dataset synthetic for a synthetic stream' % data_dir)


def _mnist(data_dir):
    tfr = os.path.join(data_dir, 'mnist.train.tfrecords')                    # hem/data/mnist.py:74-77
    if os.path.exists(tfr):
        return tfrecord.load_image_tfrecords(tfr, (28, 28, 1))
    gz = os.path.join(data_dir, 'train-images-idx3-ubyte.gz')
    if os.path.exists(gz):                                                   # hem/data/mnist.py:52-58
        with gzip.open(gz) as f:
            data = f.read()
        _, n, r, c = struct.unpack('>iiii', data[:16])
        return np.frombuffer(data[16:], dtype=np.uint8).reshape(n, r, c, 1)
    raise FileNotFoundError('no MNIST data under %s; use --dataset synthetic' % data_dir)


def get_dataset(args, sess):
    name, B = args.dataset, args.batch_size
    if name == 'synthetic' and args.model == 'pix2pix':
        return SyntheticPairSource(4, B, sess.device, 256, 1234, sess.rank), 4 * B * sess.world_size, (256, 256, 3)
    if name == 'synthetic':
        shape = (32, 32, 3)
        if args.resize:
            shape = (args.resize[1], args.resize[0], 3)
        if args.grayscale:
            shape = shape[:2] + (1,)
        return SyntheticSource(50000, shape, B, sess.device, 1234, sess.rank), 50000, shape
    if name == 'cifar':
        imgs = _cifar(args.data_dir)
    elif name == 'mnist':
        imgs = _mnist(args.data_dir)
    else:
        raise NotImplementedError('dataset %r needs an image decoder that this build does not ship '
                                  '(floorplans/nyuv2 are PNG/JPEG TFRecords)' % name)
    n = imgs.shape[0]
    # per-replica shard: rank r takes every world_size-th batch-sized block (== ops/input.py:24 on a shuffled stream)
    x = torch.from_numpy(np.ascontiguousarray(imgs)).to(sess.device).float() / 255.0
    if x.shape[1] == 28:                                  # MNIST: pad 28 -> 32 so the 2x deconv ladder fits (App. C-1)
        x = F.pad(x.permute(0, 3, 1, 2), (2, 2, 2, 2)).permute(0, 2, 3, 1)
    if args.resize:
        w, h = args.resize
        x = F.interpolate(x.permute(0, 3, 1, 2), size=(h, w), mode='bilinear', align_corners=False).permute(0, 2, 3, 1)
    if args.grayscale and x.shape[-1] == 3:
        x = (x * torch.tensor([0.2989, 0.5870, 0.1140], device=x.device)).sum(-1, keepdim=True)
    x = x.contiguous()
    shard = x[sess.rank::sess.world_size] if sess.world_size > 1 else x
    src = ArraySource(shard.cpu().numpy(), B, sess.device, shuffle_seed=(args.seed or 0) + sess.rank if args.shuffle else None)
    return src, n, tuple(x.shape[1:])
