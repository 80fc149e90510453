"""The streaming input pipeline (SURVEY section 8f-1 as the survey wrote it; VERDICT r3 missing 4): the reference's
`d.cache().repeat().shuffle(buffer_size).batch(batch_size * n_gpus)` (data.py:54-58, train.py:171-174) with each tower taking
its rows of the batch (ops/input.py:11-25).  Host work (native shuffle-buffer draw + row gather, include/tdg.h) -- no GPU."""
import ctypes
import importlib

import numpy as np
import torch

_lib = importlib.import_module('3dgan_amd._lib')
data = importlib.import_module('3dgan_amd.data')


def _draw(n_total, buffer_size, count, seed=1):
    lib = _lib.load()
    buf = (np.arange(buffer_size, dtype=np.int64) % n_total).copy()
    nxt = np.array([buffer_size % n_total], dtype=np.int64)
    state = np.random.SeedSequence([seed]).generate_state(4, dtype=np.uint64).copy()
    out = np.empty(count, dtype=np.int64)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    assert lib.tdg_shuffle_draw(p(buf), buffer_size, p(state), p(nxt), n_total, count, p(out)) == 0
    return out, buf, int(nxt[0])


def test_shuffle_buffer_has_tf_data_semantics():
    """With a stream that never repeats inside the window (n_total >> draws): every stream position is emitted at most once,
    position p cannot be emitted before draw p - buffer_size (it has not entered the buffer yet), everything emitted or still
    in the buffer is exactly the prefix of the stream that has been read, and the waiting time in the buffer is geometric
    with mean buffer_size (tf.data's ShuffleDataset: random slot out, next element in)."""
    n_total, bs, count = 10_000_000, 1000, 60000
    out, buf, nxt = _draw(n_total, bs, count)
    assert len(np.unique(out)) == count
    t = np.arange(count)
    assert np.all(out <= t + bs - 1 + 1)                       # position p is in the buffer from draw max(0, p - bs + 1) on
    assert nxt == bs + count
    assert np.array_equal(np.sort(np.concatenate([out, buf])), np.arange(bs + count))
    wait = t - np.maximum(out - bs + 1, 0)                      # draws spent in the buffer
    assert abs(wait[5 * bs:].mean() / bs - 1.0) < 0.05          # geometric, mean ~ buffer_size
    # buffer_size 1 is the identity; equal seeds give equal sequences, different seeds different ones
    ident, _, _ = _draw(50, 1, 120)
    assert np.array_equal(ident, np.arange(120) % 50)
    a, _, _ = _draw(1000, 100, 500, seed=3)
    b, _, _ = _draw(1000, 100, 500, seed=3)
    c, _, _ = _draw(1000, 100, 500, seed=4)
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_repeat_before_shuffle_mixes_epochs_and_stays_uniform():
    """repeat() sits in FRONT of shuffle() in the reference, so the buffer straddles epoch boundaries: over many epochs every
    example is drawn equally often (+- sampling noise) even though a single window of N draws is not a permutation."""
    n_total, bs = 500, 100
    out, _, _ = _draw(n_total, bs, 200 * n_total, seed=7)
    counts = np.bincount(out, minlength=n_total)
    assert counts.min() >= 198 and counts.max() <= 202         # each example leaves the buffer once per pass of the stream (+- the tail)
    first = out[:n_total]
    assert len(np.unique(first)) < n_total                     # a window of N draws repeats some examples: not an epoch permutation


def test_streaming_source_serves_each_towers_rows_of_the_global_batch():
    rng = np.random.default_rng(0)
    imgs = rng.integers(0, 256, (300, 8, 8, 3), dtype=np.uint8)
    B, world = 16, 2
    srcs = [data.StreamingSource(imgs, B, 'cpu', buffer_size=64, seed=5, rank=r, world=world) for r in range(world)]
    ref, _, _ = None, None, None
    lib_draws = None
    for step in range(12):
        got = [s.next_batch(return_indices=True) for s in srcs]
        for r, (x, idx) in enumerate(got):
            assert x.dtype == torch.float32 and tuple(x.shape) == (B, 8, 8, 3)
            assert np.array_equal(x.numpy(), imgs[idx].astype(np.float32) * np.float32(1.0 / 255.0))
        # the two towers hold DIFFERENT rows of one global batch: re-derive it from the first tower's generator state
        glob = np.concatenate([got[0][1], got[1][1]])
        assert len(glob) == B * world
    # both ranks draw the same global sequence: rank 1's rows are what rank 0 skipped
    one = data.StreamingSource(imgs, B * world, 'cpu', buffer_size=64, seed=5, rank=0, world=1)
    two = [data.StreamingSource(imgs, B, 'cpu', buffer_size=64, seed=5, rank=r, world=world) for r in range(world)]
    for step in range(5):
        _, g = one.next_batch(return_indices=True)
        parts = [s.next_batch(return_indices=True)[1] for s in two]
        assert np.array_equal(g, np.concatenate(parts))
    for s in srcs + two + [one]:
        s.close()


def test_streaming_source_without_shuffle_is_the_repeated_stream():
    imgs = np.arange(10 * 4, dtype=np.float32).reshape(10, 2, 2, 1) / 40.0
    s = data.StreamingSource(imgs, 4, 'cpu', shuffle=False)
    seen = np.concatenate([s.next_batch(return_indices=True)[1] for _ in range(6)])
    assert np.array_equal(seen, np.arange(24) % 10)
    s.close()
