"""JPEG files -> numpy, for the dataset plugins (floorplan records hold the raw bytes of the source files:
data/floorplan_tfrecords.py:26-41; the reference decodes them with `tf.image.decode_image(..., channels=3)`, data.py:15).
The decoder is the native `tdg_jpeg_decode` of lib3dgan_hip.so (3dgan_amd/csrc/tdg_host.cpp): baseline / extended-sequential
Huffman files, libjpeg's default arithmetic.  Host work only."""
import ctypes as C

import numpy as np

from . import _lib


def is_jpeg(data):
    return len(data) >= 3 and data[0] == 0xFF and data[1] == 0xD8 and data[2] == 0xFF


def decode(data, channels=3):
    """uint8 [height, width, 3] (grayscale files are replicated to three channels, as decode_image(channels=3) does)."""
    if channels not in (0, 3):
        raise ValueError('jpeg.decode produces 3 channels (channels=%r asked for)' % (channels,))
    data = bytes(data)
    w, h, c = C.c_int(0), C.c_int(0), C.c_int(0)
    _lib.call('tdg_jpeg_info', data, len(data), C.byref(w), C.byref(h), C.byref(c))
    out = np.empty((h.value, w.value, 3), np.uint8)
    _lib.call('tdg_jpeg_decode', data, len(data), out.ctypes.data, out.nbytes)
    return out
