"""PNG decoding for the input pipeline (the reference: `tf.image.decode_png` / `decode_image`,
hem/data/nyuv2.py:152-153, data.py:15).  Container parsing and the IDAT inflate are Python (struct + zlib); the
scanline reconstruction -- the only per-byte work -- is the native `tdg_png_unfilter` of lib3dgan_hip.so.

Supported: non-interlaced, bit depth 8 or 16, colour types 0 (grey), 2 (RGB), 4 (grey + alpha), 6 (RGBA): what the
dataset converters write (`*_i.png` 8-bit RGB, `*_f.png` 16-bit grey depth, hem/data/nyuv2.py:128-131).  Palette images,
1/2/4-bit depths and Adam7 interlacing raise ValueError; JPEG (the floorplan records are whatever the source files
were: `decode_image`) is reported as such here -- 3dgan_amd/jpeg.py decodes those.
"""
import struct
import zlib

import numpy as np

from . import _lib

_SIG = b'\x89PNG\r\n\x1a\n'
_CHANNELS = {0: 1, 2: 3, 4: 2, 6: 4}


def sniff(data):
    if data[:8] == _SIG:
        return 'png'
    if data[:2] == b'\xff\xd8':
        return 'jpeg'
    if data[:6] in (b'GIF87a', b'GIF89a'):
        return 'gif'
    if data[:2] == b'BM':
        return 'bmp'
    return 'unknown'


def decode(data, channels=0):
    """-> uint8 or uint16 array [H, W, C].  `channels` = 1 or 3 converts like tf.image.decode_png(channels=...)
    (grey -> RGB by replication, RGB -> grey by the ITU-R 601 weights, alpha dropped)."""
    kind = sniff(data)
    if kind != 'png':
        raise ValueError('image is %s, not PNG: this build decodes PNG records only' % kind)
    pos, idat, hdr = 8, [], None
    while pos + 8 <= len(data):
        n, tag = struct.unpack('>I4s', data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if len(body) != n:
            raise ValueError('PNG chunk %r is truncated' % tag)
        crc, = struct.unpack('>I', data[pos + 8 + n:pos + 12 + n])
        if zlib.crc32(tag + body) & 0xffffffff != crc:
            raise ValueError('PNG chunk %r fails its CRC' % tag)
        if tag == b'IHDR':
            hdr = struct.unpack('>IIBBBBB', body)
        elif tag == b'IDAT':
            idat.append(body)
        elif tag == b'IEND':
            break
        pos += 12 + n
    if hdr is None or not idat:
        raise ValueError('PNG without IHDR / IDAT')
    w, h, depth, color, comp, flt, interlace = hdr
    if color not in _CHANNELS or depth not in (8, 16) or interlace != 0 or comp != 0 or flt != 0:
        raise ValueError('unsupported PNG: colour type %d, bit depth %d, interlace %d' % (color, depth, interlace))
    c = _CHANNELS[color]
    bpp = c * depth // 8
    row_bytes = w * bpp
    raw = zlib.decompress(b''.join(idat))
    if len(raw) != h * (row_bytes + 1):
        raise ValueError('PNG data is %d bytes, expected %d' % (len(raw), h * (row_bytes + 1)))
    out = np.empty(h * row_bytes, dtype=np.uint8)
    _lib.call('tdg_png_unfilter', raw, h, row_bytes, bpp, out.ctypes.data)
    if depth == 16:
        img = out.view('>u2').astype(np.uint16).reshape(h, w, c)
    else:
        img = out.reshape(h, w, c)
    if color in (4, 6):
        img = img[..., :-1]                       # drop alpha
    if channels == 3 and img.shape[-1] == 1:
        img = np.repeat(img, 3, axis=-1)
    elif channels == 1 and img.shape[-1] == 3:
        f = img.astype(np.float64) @ np.array([0.299, 0.587, 0.114])
        img = np.rint(f).astype(img.dtype)[..., None]
    return img
