"""ASan / UBSan run of the native JPEG decoder on mutated files (CPU only, diagnostics).
build:  g++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -shared 3dgan_amd/csrc/tdg_host.cpp STUB.cpp -I3dgan_amd/csrc -o /tmp/libtdg_host_asan.so
        (STUB.cpp defines tdg_set_error(const char*, ...)); run with LD_PRELOAD=libasan.so:libubsan.so ASAN_OPTIONS=detect_leaks=0
round 4: 9000 mutated / truncated files (4:4:4, 4:2:2, 4:2:0, grayscale): 2402 decoded, 6598 refused, no sanitizer report."""
import ctypes as C, io, struct, sys
import numpy as np
from PIL import Image
lib = C.CDLL('/tmp/libtdg_host_asan.so')
lib.tdg_jpeg_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.tdg_jpeg_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
def dec(data):
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    if lib.tdg_jpeg_info(data, len(data), w, h, c): return None
    out = np.empty((h.value, w.value, 3), np.uint8)
    if lib.tdg_jpeg_decode(data, len(data), out.ctypes.data, out.nbytes): return None
    return out
rng = np.random.default_rng(0)
n_ok = n_err = 0
for sub in (0, 1, 2):
    for gray in (False, True):
        a = rng.integers(0, 256, (41, 67, 3), dtype=np.uint8)
        b = io.BytesIO(); Image.fromarray(a[..., 0] if gray else a).save(b, 'JPEG', quality=70, subsampling=sub); data = b.getvalue()
        assert dec(data) is not None
        for t in range(1500):
            bad = bytearray(data)
            for _ in range(int(rng.integers(1, 6))):
                bad[int(rng.integers(2, len(bad)))] = int(rng.integers(0, 256))
            if t % 5 == 0: bad = bad[:int(rng.integers(4, len(bad)))]
            r = dec(bytes(bad)); n_ok += r is not None; n_err += r is None
big = bytes([0xff, 0xd8, 0xff, 0xc4]) + struct.pack('>H', 2 + 17 + 255) + bytes([0x00, 255] + [0] * 15) + bytes(range(255)) + data[2:]
assert dec(big) is None
print('asan/ubsan run: %d decoded, %d refused, no report' % (n_ok, n_err))
