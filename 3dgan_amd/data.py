"""Per-replica input sources (`x` of `gan(x, args)`): anything with `.next_batch()` returning a
device float32 tensor [B, H, W, C] in [0, 1] (the value range of the reference's parsers,
data.py:22,30).  The reference's TFRecord pipeline (data.py:34-60) is host I/O outside the hot
path; `SyntheticSource` is the bench / smoke input of section 8d.
"""
import numpy as np
import torch


class SyntheticSource:
    """uint8 U{0..255} images / 255, resident in HBM, served as consecutive B-image batches.
    Rank r of n starts at a different offset, mirroring the row slices of ops/input.py:24."""

    def __init__(self, n_images, image_shape, batch_size, device, seed=1234, rank=0):
        h, w, c = image_shape
        n_images = max(n_images, batch_size)
        rng = np.random.default_rng(seed + 7919 * rank)
        pool = rng.integers(0, 256, size=(n_images, h, w, c), dtype=np.uint8)
        self.pool = (torch.from_numpy(pool).to(device).float() / 255.0).contiguous()
        self.B, self.n, self.i = batch_size, n_images, 0

    def next_batch(self):
        if self.i + self.B > self.n:
            self.i = 0
        b = self.pool[self.i:self.i + self.B]
        self.i += self.B
        return b


class ArraySource:
    """Batches from a host array [N, H, W, C] in [0,1] (tests, small real datasets)."""

    def __init__(self, array, batch_size, device, shuffle_seed=None):
        self.data = torch.as_tensor(np.asarray(array), dtype=torch.float32).to(device).contiguous()
        self.B, self.i = batch_size, 0
        self.gen = None if shuffle_seed is None else torch.Generator(device='cpu').manual_seed(shuffle_seed)
        self.perm = None

    def next_batch(self):
        n = self.data.shape[0]
        if self.i + self.B > n or (self.gen is not None and self.perm is None):
            self.i = 0
            if self.gen is not None:
                self.perm = torch.randperm(n, generator=self.gen).to(self.data.device)
        idx = slice(self.i, self.i + self.B)
        self.i += self.B
        return (self.data[self.perm[idx]] if self.perm is not None else self.data[idx]).contiguous()


class SyntheticPairSource:
    """(rgb, depth) pairs for pix2pix: x ~ U[0,1) [B,256,256,3], y ~ U(0,1) exclusive of
    exact 0/1 [B,256,256,1] (the nyuv2 plugin drops depth maps containing 0 or 1, hem/data/nyuv2.py:258-266)."""

    def __init__(self, n_batches, batch_size, device, size=256, seed=1234, rank=0):
        g = torch.Generator(device='cpu').manual_seed(seed + 7919 * rank)
        self.x = torch.rand(n_batches, batch_size, size, size, 3, generator=g).to(device)
        self.y = (torch.rand(n_batches, batch_size, size, size, 1, generator=g) * 0.98 + 0.01).to(device)
        self.i = 0

    def next_batch(self):
        k = self.i % self.x.shape[0]
        self.i += 1
        return self.x[k], self.y[k]
