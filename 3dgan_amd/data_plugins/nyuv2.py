"""NYU Depth v2 RGB -> depth pairs (hem/data/nyuv2.py), the input of `--model pix2pix` (examples/pix2pix.config).

Records (`nyuv2.train.tfrecords`, written by hem/data/nyuv2.py:121-146): `image` = PNG bytes of the 8-bit RGB frame,
`depth` = PNG bytes of the 16-bit depth frame, plus width / height / channels / filenames.  Parsing as in
`parse_tfrecord` (:148-246): decode_png; optional `--resize W H` (TF-1.x bilinear) or `--random_crop H W` (one crop
window per draw, the same for image and depth, :206-208); image / 255, depth / 65535; pairs whose depth crop contains an
exact 0 or 1 are dropped (`ignore_incomplete_depthmaps`, :258-262).  Output per batch: (x [B, H, W, 3], y [B, H, W, 1])
in [0, 1], NHWC, on the device.

`--include_location`, `--normalize`, `--include_originals` add tensors that only the thesis samplers read
(hem/models/paper_*.py, out of scope): they are accepted and ignored with a warning.
"""
import os
import sys

import numpy as np
import torch

from .DataPlugin import DataPlugin, find_file, dataset_dirs, write_cache_atomically
from ._common import resize_bilinear_tf1
from .. import tfrecord, png

_dataset_files = {'train': 'nyuv2.train.tfrecords', 'validate': 'nyuv2.validate.tfrecords', 'test': 'nyuv2.test.tfrecords'}


class PairSource:
    """Device-resident decoded frames; every next_batch() draws B frames (shuffled epoch order, per-replica shard), one
    random crop window per frame, and re-draws windows whose depth crop holds an exact 0 or 1."""

    def __init__(self, rgb_u8, depth_u16, batch_size, device, crop, resize, seed, rank, world):
        self.rgb = torch.from_numpy(rgb_u8[rank::world] if world > 1 else rgb_u8).to(device)            # [N, H, W, 3] uint8
        self.depth = torch.from_numpy((depth_u16[rank::world] if world > 1 else depth_u16).astype(np.int32)).to(device)
        self.B, self.device, self.crop, self.resize = batch_size, device, crop, resize
        self.gen = torch.Generator(device='cpu').manual_seed(int(seed) + 7919 * rank)
        self.n = self.rgb.shape[0]
        self.barren_passes = 8                       # give up after this many passes over the shard without ONE valid crop
        if not crop:
            # without a random window a frame's verdict never changes: decide once, then draw only frames that pass
            # (the reference's filter is unbounded, hem/data/nyuv2.py:258-266; re-testing them every epoch is not)
            keep = []
            for i0 in range(0, self.n, 64):
                idx = torch.arange(i0, min(i0 + 64, self.n))
                keep.append(idx[self._complete(self._make(idx)[1]).cpu()])
            keep = torch.cat(keep) if keep else torch.zeros(0, dtype=torch.long)
            if keep.numel() == 0:
                raise RuntimeError('nyuv2: every one of the %d depth maps has sensor gaps (an exact 0 or 1)' % self.n)
            self.rgb, self.depth = self.rgb[keep.to(device)], self.depth[keep.to(device)]
            self.n = int(keep.numel())
        self.perm, self.i = torch.randperm(self.n, generator=self.gen), 0

    def _indices(self, k):
        out = []
        while len(out) < k:
            if self.i >= self.n:
                self.perm, self.i = torch.randperm(self.n, generator=self.gen), 0
            take = min(k - len(out), self.n - self.i)
            out.append(self.perm[self.i:self.i + take])
            self.i += take
        return torch.cat(out)

    def _make(self, idx):
        rgb, depth = self.rgb[idx.to(self.device)], self.depth[idx.to(self.device)]
        x = rgb.float()
        y = depth.float()[..., None]
        if self.resize:
            w, h = self.resize
            x, y = resize_bilinear_tf1(x, h, w), resize_bilinear_tf1(y, h, w)
        if self.crop:
            ch, cw = self.crop
            H, W = x.shape[1], x.shape[2]
            top = torch.randint(0, H - ch + 1, (len(idx),), generator=self.gen).to(self.device)
            left = torch.randint(0, W - cw + 1, (len(idx),), generator=self.gen).to(self.device)
            rows = (top[:, None] + torch.arange(ch, device=self.device)[None])                  # [k, ch]
            cols = (left[:, None] + torch.arange(cw, device=self.device)[None])                 # [k, cw]
            bi = torch.arange(len(idx), device=self.device)[:, None, None]
            x = x[bi, rows[:, :, None], cols[:, None, :]]
            y = y[bi, rows[:, :, None], cols[:, None, :]]
        return x / 255.0, y / 65535.0

    @staticmethod
    def _complete(y):
        flat = y.reshape(y.shape[0], -1)
        return ~((flat == 0).any(1) | (flat == 1).any(1))                     # hem/data/nyuv2.py:258-262

    def next_batch(self):
        """Frames are redrawn until B crops pass the filter, however sparse the valid ones are (the reference's filter is
        unbounded); the only failure is NO progress: `barren_passes` whole passes over the shard without a single valid crop."""
        xs, ys, need, barren = [], [], self.B, 0
        while need > 0:
            x, y = self._make(self._indices(need))
            ok = self._complete(y) if self.crop else torch.ones(y.shape[0], dtype=torch.bool, device=y.device)
            if ok.any():
                xs.append(x[ok])
                ys.append(y[ok])
                need -= int(ok.sum())
                barren = 0
            else:
                barren += int(y.shape[0])
                if barren >= self.barren_passes * max(self.n, 1):
                    raise RuntimeError('nyuv2: no depth crop without sensor gaps in %d passes over %d frames' % (self.barren_passes, self.n))
        return torch.cat(xs)[:self.B].contiguous(), torch.cat(ys)[:self.B].contiguous()


class NYUv2Dataset(DataPlugin):
    name = 'nyuv2'

    @staticmethod
    def arguments():
        """hem/data/nyuv2.py:40-78."""
        return {
            '--resize': {'type': int, 'nargs': 2, 'help': 'Resize input images to size w x h.'},
            '--random_crop': {'type': int, 'nargs': 2, 'help': 'Randomly crop the input images to size h x w.'},
            '--include_location': {'action': 'store_true', 'default': False,
                                   'help': 'Thesis samplers only: accepted, ignored.'},
            '--skip_invalid': {'action': 'store_true', 'default': False,
                               'help': 'Parsed but unused by the reference; depth maps with gaps are always dropped.'},
            '--normalize': {'action': 'store_true', 'default': False, 'help': 'Thesis samplers only: accepted, ignored.'},
            '--include_originals': {'type': int, 'nargs': 2, 'help': 'Thesis samplers only: accepted, ignored.'},
        }

    @staticmethod
    def check_prepared_datasets(storage_dir):
        return DataPlugin.check_files(storage_dir, list(_dataset_files.values()))

    @staticmethod
    def load(args, split='train'):
        cache = os.path.join(args.cache_dir, 'nyuv2.%s.npz' % split) if getattr(args, 'cache_dir', None) else None
        if cache and os.path.exists(cache):
            z = np.load(cache)
            return z['rgb'], z['depth']
        tfr = find_file(args, [_dataset_files[split]])
        if not tfr:
            raise FileNotFoundError('no %s under %s; use --dataset synthetic' % (_dataset_files[split], dataset_dirs(args)))
        rgb, depth = [], []
        for rec in tfrecord.read_records(tfr):
            ex = tfrecord.parse_example(rec)
            rgb.append(png.decode(ex['image'], channels=3))                 # hem/data/nyuv2.py:152
            d = png.decode(ex['depth'], channels=1)                         # :153 (dtype uint16)
            depth.append(d[..., 0].astype(np.uint16) if d.dtype != np.uint16 else d[..., 0])
        rgb, depth = np.stack(rgb), np.stack(depth)
        if cache:
            write_cache_atomically(cache, lambda f: np.savez(f, rgb=rgb, depth=depth))
        return rgb, depth

    @staticmethod
    def get_source(args, sess):
        for flag in ('include_location', 'normalize', 'include_originals'):
            if getattr(args, flag, None):
                sys.stderr.write('WARNING: --%s only feeds the thesis samplers; ignored\n' % flag)
        rgb, depth = NYUv2Dataset.load(args)
        crop = tuple(args.random_crop) if getattr(args, 'random_crop', None) else None
        seed = args.seed if isinstance(getattr(args, 'seed', None), int) else 0
        src = PairSource(rgb, depth, args.batch_size, sess.device, crop, getattr(args, 'resize', None), seed, sess.rank,
                         sess.world_size)
        h, w = crop if crop else ((args.resize[1], args.resize[0]) if getattr(args, 'resize', None) else rgb.shape[1:3])
        return src, rgb.shape[0], (h, w, 3)
