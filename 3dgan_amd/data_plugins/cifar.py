"""CIFAR-10 (hem/data/cifar.py; gen-1 data.py:26-31,38-40).  The reference's parsers are broken in both generations
(SURVEY.md App. C-6), so this reads what the converter actually wrote -- 3072 HWC uint8 bytes under key `image`
(data/cifar_tfrecords.py:27-32) -- or the original python-pickle batches."""
import os
import pickle

import numpy as np

from .DataPlugin import DataPlugin, find_file, dataset_dirs
from ._common import finish_images
from .. import tfrecord


class CifarDataset(DataPlugin):
    name = 'cifar'

    @staticmethod
    def arguments():
        return {'--resize': {'type': int, 'nargs': 2, 'help': 'Resize input images to size w x h.'}}

    @staticmethod
    def check_prepared_datasets(storage_dir):
        return DataPlugin.check_files(storage_dir, ['cifar.32.train.tfrecords'])

    @staticmethod
    def load(args):
        tfr = find_file(args, ['cifar.32.train.tfrecords'])                   # data.py:39
        if tfr:
            return tfrecord.load_image_tfrecords(tfr, (32, 32, 3))
        pk = find_file(args, ['cifar-10-batches-py'])
        if pk and os.path.isdir(pk):                                          # data/cifar_tfrecords.py:23-29
            out = []
            for i in range(1, 6):
                with open(os.path.join(pk, 'data_batch_%d' % i), 'rb') as f:
                    d = pickle.load(f, encoding='bytes')
                out.append(d[b'data'].reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1))
            return np.concatenate(out)
        raise FileNotFoundError('no CIFAR-10 data under %s (expected cifar.32.train.tfrecords or cifar-10-batches-py/); '
                                'use --dataset synthetic for a synthetic stream' % dataset_dirs(args))

    @staticmethod
    def get_source(args, sess):
        return finish_images(CifarDataset.load(args), args, sess)
