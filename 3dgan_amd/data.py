"""Per-replica input sources (`x` of `gan(x, args)`): anything with `.next_batch()` returning a
device float32 tensor [B, H, W, C] in [0, 1] (the value range of the reference's parsers,
data.py:22,30).  The reference's TFRecord pipeline (data.py:34-60) is host I/O outside the hot
path; `SyntheticSource` is the bench / smoke input of section 8d.
"""
import ctypes
import queue
import threading

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


class StreamingSource:
    """The reference's host pipeline for a dataset that stays in HOST memory (data.py:54-58, train.py:171-174):
    cache -> repeat() -> shuffle(buffer_size) -> batch(batch_size * n_gpus), each tower taking its rows of the batch
    (ops/input.py:11-25).  Here: the cached examples are a host array [N, H, W, C] (uint8 or float32; an np.memmap works);
    a producer thread draws example INDICES from a shuffle buffer with tf.data's semantics (`tdg_shuffle_draw`: uniformly
    random slot out, next stream element in, the stream repeating), assembles this replica's rows of the global batch in a
    ring of PINNED host buffers (`tdg_gather_rows`; ctypes releases the GIL) and issues the host-to-device copies on a side
    stream; `next_batch()` makes the compute stream wait for the copy's event, converts to float32 in [0, 1] on the device
    and returns.  Every rank draws the SAME global index sequence (one seed) and takes rows [rank B, (rank + 1) B) of it.
    `post`: an optional device-side transform of the float batch (resize / grayscale of the image plugins)."""

    def __init__(self, array, batch_size, device, buffer_size=10000, seed=0, rank=0, world=1, ring=3, shuffle=True, post=None):
        from . import _lib
        self.lib = _lib.load()
        a = np.asarray(array) if not isinstance(array, np.memmap) else array
        if a.dtype not in (np.uint8, np.float32) or not a.flags['C_CONTIGUOUS']:
            a = np.ascontiguousarray(a, dtype=np.float32 if a.dtype.kind == 'f' else np.uint8)
        self.data, self.n = a, int(a.shape[0])
        self.row_shape = tuple(a.shape[1:])
        self.row_bytes = int(np.prod(self.row_shape)) * a.dtype.itemsize
        self.B, self.world, self.rank, self.device, self.post = batch_size, world, rank, torch.device(device), post
        self.scale = 1.0 / 255.0 if a.dtype == np.uint8 else 1.0
        self.shuffle = shuffle
        # shuffle-buffer state: the first buffer_size elements of the REPEATED stream, as tf.data fills it
        blen = max(1, int(buffer_size)) if shuffle else 1
        self.buf = (np.arange(blen, dtype=np.int64) % self.n).copy()
        self.next_in = np.array([blen % self.n if shuffle else 0], dtype=np.int64)
        sm = np.random.SeedSequence([int(seed), 0x3d6a]).generate_state(4, dtype=np.uint64)
        self.state = np.array(sm, dtype=np.uint64)
        self.draws = np.empty(self.B * self.world, dtype=np.int64)
        cuda = self.device.type == 'cuda'
        tdt = torch.uint8 if a.dtype == np.uint8 else torch.float32
        self.host = [torch.empty((self.B,) + self.row_shape, dtype=tdt, pin_memory=cuda) for _ in range(ring)]
        self.dev = [torch.empty((self.B,) + self.row_shape, dtype=tdt, device=self.device) for _ in range(ring)] if cuda else self.host
        self.copy_stream = torch.cuda.Stream(device=self.device) if cuda else None
        self.ready = [torch.cuda.Event() for _ in range(ring)] if cuda else None
        self.done = [torch.cuda.Event() for _ in range(ring)] if cuda else None
        self.used = [False] * ring
        self.q = queue.Queue(maxsize=ring - 1)           # slots whose copy has been issued, in order
        self.free = queue.Queue()
        for k in range(ring):
            self.free.put(k)
        self.error = None
        self.stop = False
        self.thread = threading.Thread(target=self._produce, daemon=True)
        self.thread.start()

    def _ptr(self, arr):
        return ctypes.c_void_p(arr.ctypes.data)

    def _next_indices(self):
        """This replica's rows of the next global batch."""
        n_draw = self.B * self.world
        if self.shuffle:
            rc = self.lib.tdg_shuffle_draw(self._ptr(self.buf), self.buf.shape[0], self._ptr(self.state), self._ptr(self.next_in),
                                           self.n, n_draw, self._ptr(self.draws))
            if rc:
                raise RuntimeError('tdg_shuffle_draw failed: %s' % self.lib.tdg_last_error().decode())
        else:
            start = int(self.next_in[0])
            self.draws[:] = (start + np.arange(n_draw, dtype=np.int64)) % self.n
            self.next_in[0] = (start + n_draw) % self.n
        return self.draws[self.rank * self.B:(self.rank + 1) * self.B]

    def _produce(self):
        try:
            while not self.stop:
                k = self.free.get()
                if k is None:
                    return
                idx = np.ascontiguousarray(self._next_indices())
                if self.done is not None and self.used[k]:
                    self.done[k].synchronize()            # the consumer has converted this slot's previous batch
                out = self.host[k].numpy()
                rc = self.lib.tdg_gather_rows(ctypes.c_void_p(self.data.ctypes.data), self.n, self.row_bytes, self._ptr(idx), self.B,
                                              ctypes.c_void_p(out.ctypes.data))
                if rc:
                    raise RuntimeError('tdg_gather_rows failed: %s' % self.lib.tdg_last_error().decode())
                if self.copy_stream is not None:
                    with torch.cuda.stream(self.copy_stream):
                        self.dev[k].copy_(self.host[k], non_blocking=True)
                        self.ready[k].record(self.copy_stream)
                self.used[k] = True
                self.q.put((k, idx.copy()))
        except BaseException as e:                        # surfaced by the next next_batch()
            self.error = e
            self.q.put((None, None))

    def next_batch(self, return_indices=False):
        k, idx = self.q.get()
        if k is None:
            raise RuntimeError('StreamingSource: the producer thread failed') from self.error
        if self.copy_stream is not None:
            torch.cuda.current_stream(self.device).wait_event(self.ready[k])
        x = self.dev[k].to(torch.float32)
        if self.scale != 1.0:
            x = x * self.scale
        if self.copy_stream is not None:
            self.done[k].record(torch.cuda.current_stream(self.device))
        else:
            x = x.clone()                                  # (CPU device: the ring slot is reused)
        self.free.put(k)
        if self.post is not None:
            x = self.post(x)
        return (x, idx) if return_indices else x

    def close(self):
        self.stop = True
        self.free.put(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
