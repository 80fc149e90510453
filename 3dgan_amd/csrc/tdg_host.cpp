// Host-side native helpers of lib3dgan_hip.so: the input pipeline's image decoding (the reference decodes with
// TensorFlow's C++ `tf.image.decode_png`, hem/data/nyuv2.py:152-153; floorplans: data.py:15).  No device code here.
#include <stdint.h>
#include <stdlib.h>

#include "../../include/tdg.h"

void tdg_set_error(const char* fmt, ...);

static inline int paeth(int a, int b, int c) {
  const int p = a + b - c;
  const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

extern "C" int tdg_png_unfilter(const unsigned char* filtered, int rows, int row_bytes, int bpp, unsigned char* out) {
  if (!filtered || !out || rows <= 0 || row_bytes <= 0 || bpp <= 0 || bpp > 8) {
    tdg_set_error("tdg_png_unfilter: bad arguments (rows %d, row_bytes %d, bpp %d)", rows, row_bytes, bpp);
    return TDG_EINVAL;
  }
  for (int r = 0; r < rows; ++r) {
    const unsigned char* in = filtered + (size_t)r * (row_bytes + 1);
    unsigned char* cur = out + (size_t)r * row_bytes;
    const unsigned char* up = r ? cur - row_bytes : nullptr;
    const int ft = in[0];
    ++in;
    switch (ft) {
      case 0:
        for (int i = 0; i < row_bytes; ++i) cur[i] = in[i];
        break;
      case 1:
        for (int i = 0; i < row_bytes; ++i) cur[i] = (unsigned char)(in[i] + (i >= bpp ? cur[i - bpp] : 0));
        break;
      case 2:
        for (int i = 0; i < row_bytes; ++i) cur[i] = (unsigned char)(in[i] + (up ? up[i] : 0));
        break;
      case 3:
        for (int i = 0; i < row_bytes; ++i)
          cur[i] = (unsigned char)(in[i] + (((i >= bpp ? cur[i - bpp] : 0) + (up ? up[i] : 0)) >> 1));
        break;
      case 4:
        for (int i = 0; i < row_bytes; ++i) {
          const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
          cur[i] = (unsigned char)(in[i] + paeth(a, b, c));
        }
        break;
      default:
        tdg_set_error("tdg_png_unfilter: scanline %d has filter type %d", r, ft);
        return TDG_EINVAL;
    }
  }
  return TDG_OK;
}
