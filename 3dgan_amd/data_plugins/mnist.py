"""MNIST (hem/data/mnist.py:52-102): `mnist.train.tfrecords` (784 raw bytes under `image`) or the idx gzip files."""
import gzip
import struct

import numpy as np

from .DataPlugin import DataPlugin, find_file, dataset_dirs
from ._common import finish_images
from .. import tfrecord


class MNISTDataset(DataPlugin):
    name = 'mnist'

    @staticmethod
    def arguments():
        return {'--resize': {'type': int, 'nargs': 2, 'help': 'Resize input images to size w x h.'}}

    @staticmethod
    def check_prepared_datasets(storage_dir):
        return DataPlugin.check_files(storage_dir, ['mnist.train.tfrecords'])

    @staticmethod
    def load(args):
        tfr = find_file(args, ['mnist.train.tfrecords'])                      # hem/data/mnist.py:74-77
        if tfr:
            return tfrecord.load_image_tfrecords(tfr, (28, 28, 1))
        gz = find_file(args, ['train-images-idx3-ubyte.gz'])
        if gz:                                                                # hem/data/mnist.py:52-58
            with gzip.open(gz) as f:
                data = f.read()
            _, n, r, c = struct.unpack('>iiii', data[:16])
            return np.frombuffer(data[16:], dtype=np.uint8).reshape(n, r, c, 1)
        raise FileNotFoundError('no MNIST data under %s; use --dataset synthetic' % dataset_dirs(args))

    @staticmethod
    def get_source(args, sess):
        return finish_images(MNISTDataset.load(args), args, sess, pad_to_32=True)
