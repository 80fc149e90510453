"""Floorplans (hem/data/floorplan.py:60-122; gen-1 data.py:6-23): `floorplans.train.tfrecords`, one
`tf.train.Example` per image with the ENCODED file bytes under `image` (+ width / height / channels / filename),
decoded, resized to 64 x 64 with TF-1.x bilinear `resize_images` and scaled to [0, 1].

Supported encodings: JPEG (3dgan_amd/jpeg.py: baseline / extended-sequential files) and PNG (3dgan_amd/png.py).  The
reference calls `tf.image.decode_image`, which also takes GIF / BMP and progressive JPEG; a record in one of those raises
naming the format.
Decoded images are cached as `<cache_dir>/floorplans.64.npy` when --cache_dir is given (`d.cache(...)`, data.py:51)."""
import os

import numpy as np
import torch

from .DataPlugin import DataPlugin, find_file, dataset_dirs, write_cache_atomically
from ._common import finish_images, resize_bilinear_tf1
from .. import tfrecord, png, jpeg


class FloorplanDataset(DataPlugin):
    name = 'floorplan'

    @staticmethod
    def arguments():
        return {'--resize': {'type': int, 'nargs': 2, 'help': 'Resize input images to size w x h (after the 64 x 64 of the parser).'}}

    @staticmethod
    def check_prepared_datasets(storage_dir):
        return DataPlugin.check_files(storage_dir, ['floorplans.train.tfrecords', 'floorplans.validate.tfrecords',
                                                    'floorplans.test.tfrecords'])

    @staticmethod
    def load(args):
        cache = os.path.join(args.cache_dir, 'floorplans.64.npy') if getattr(args, 'cache_dir', None) else None
        if cache and os.path.exists(cache):
            return np.load(cache)
        tfr = find_file(args, ['floorplans.train.tfrecords'])
        if not tfr:
            raise FileNotFoundError('no floorplans.train.tfrecords under %s; use --dataset synthetic' % dataset_dirs(args))
        out = []
        for rec in tfrecord.read_records(tfr):
            ex = tfrecord.parse_example(rec)
            raw = ex['image']
            img = jpeg.decode(raw) if jpeg.is_jpeg(raw) else png.decode(raw, channels=3)      # decode_image(channels=3), data.py:15
            x = torch.from_numpy(img.astype(np.float32))[None]
            x = resize_bilinear_tf1(x, 64, 64)                                # data.py:21
            out.append(np.clip(np.rint(x[0].numpy()), 0, 255).astype(np.uint8))
        imgs = np.stack(out)
        if cache:
            write_cache_atomically(cache, lambda f: np.save(f, imgs))
        return imgs

    @staticmethod
    def get_source(args, sess):
        return finish_images(FloorplanDataset.load(args), args, sess)
