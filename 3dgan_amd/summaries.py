"""Summaries-lite: what a person looks at to compare a run with the reference -- the loss scalars
and the 8x8 `inputs` / `fake` montages of `models/gan.py:93-107` -- without TensorBoard: one CSV row per epoch
and two PNG files per epoch, written with zlib only.

montage(): the tensor algebra of `ops/summaries.py:97-124` (`montage_summary`): split the batch into n chunks,
stack the chunks vertically, then lay the m images of every chunk out horizontally, i.e. image j*m + r lands at
block row j, block column r.  factorization(): `ops/summaries.py:79-92`.
"""
import os
import struct
import zlib
from math import sqrt

import numpy as np


def factorization(n):
    """ops/summaries.py:79-92: the largest i <= sqrt(n) dividing n, as (i, n // i)."""
    for i in range(int(sqrt(float(n))), 0, -1):
        if n % i == 0:
            return (i, int(n / i))


def montage(x, m=0, n=0):
    """x: [m*n, H, W, C] (or [m*n, H, W]) -> [n*H, m*W, C]."""
    x = np.asarray(x)
    if n == 0 or m == 0:
        m, n = factorization(x.shape[0])
    if x.shape[0] != m * n:
        raise ValueError('montage: %d images do not fill a %d x %d grid' % (x.shape[0], m, n))
    chunks = np.split(x, n, axis=0)                     # n x [m, H, W, C]
    tall = np.concatenate(chunks, axis=1)               # [m, n*H, W, C]
    out = np.concatenate(list(tall), axis=1)            # [n*H, m*W, C]
    if out.ndim < 3:
        out = out[:, :, None]
    return out


def png_bytes(img01):
    """8-bit PNG (grey, RGB or RGBA by channel count) of an array [H, W, C] in [0, 1]; filter type 0 on every row."""
    a = np.asarray(img01, dtype=np.float64)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    if c not in (1, 3, 4):
        raise ValueError('png: %d channels' % c)
    u8 = np.clip(np.rint(a * 255.0), 0, 255).astype(np.uint8)
    raw = b''.join(b'\x00' + u8[r].tobytes() for r in range(h))

    def chunk(tag, data):
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)
    color = {1: 0, 3: 2, 4: 6}[c]
    return (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, color, 0, 0, 0)) +
            chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


def write_png(path, img01):
    with open(path, 'wb') as f:
        f.write(png_bytes(img01))


def write_epoch(directory, epoch, status, real_pm1=None, fake_pm1=None, examples=64):
    """One `losses.csv` row and, when samples are given ([-1,1], NHWC), the two montages of models/gan.py:99-103
    (rescaled to [0,1]; an 8 x 8 grid when there are 64 examples, else the reference's factorization)."""
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, 'losses.csv')
    keys = sorted(status)
    new = not os.path.exists(path)
    with open(path, 'a') as f:
        if new:
            f.write('epoch,' + ','.join(keys) + '\n')
        f.write('%d,' % epoch + ','.join('%.8g' % float(status[k]) for k in keys) + '\n')
    written = [path]
    for name, t in (('inputs', real_pm1), ('fake', fake_pm1)):
        if t is None:
            continue
        a = (np.asarray(t, dtype=np.float32)[:examples] + 1.0) / 2.0
        m, n = (8, 8) if a.shape[0] == 64 else factorization(a.shape[0])
        p = os.path.join(directory, 'montage-%s-%04d.png' % (name, epoch))
        write_png(p, montage(a, m, n))
        written.append(p)
    return written
