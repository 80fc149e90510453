// Host-side native helpers of lib3dgan_hip.so: the input pipeline's image decoding (the reference decodes with
// TensorFlow's C++ `tf.image.decode_png`, hem/data/nyuv2.py:152-153; floorplans: data.py:15).  No device code here.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

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

// ---------------------------------------------------------------------------------------------------------------------
// Baseline / extended-sequential JPEG (8 bit, Huffman, 1 or 3 components, sampling factors 1 and 2, restart intervals):
// what `tf.image.decode_image(..., channels=3)` does for the reference's floorplan records, whose `image` feature holds
// the raw bytes of the source files (data/floorplan_tfrecords.py:26-41, data.py:15).  The arithmetic follows the decoder
// TensorFlow links (libjpeg's defaults): the slow-but-accurate integer inverse DCT, "fancy" (triangle) chroma upsampling
// for 2:1 factors, 16-bit fixed-point YCbCr -> RGB.  Progressive / arithmetic-coded / 12-bit / CMYK files are refused by
// name.
namespace {

struct JHuff {
  unsigned char bits[17] = {0};
  unsigned char vals[256] = {0};
  int mincode[18], maxcode[18], valptr[17];
  short look[512];                 // 9-bit lookahead: (length << 8) | value, 0 = longer than 9 bits
  bool present = false;
  // false: the code-length counts do not form a prefix code (more codes of some length than that length has left:
  // libjpeg's "bogus Huffman table"); such a table would index past the 9-bit lookahead
  bool build() {
    int code = 0, k = 0;
    present = false;
    for (int l = 1; l <= 16; ++l) {
      valptr[l] = k;
      mincode[l] = code;
      code += bits[l];
      if (code > (1 << l)) return false;
      k += bits[l];
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    for (int i = 0; i < 512; ++i) look[i] = 0;
    code = 0; k = 0;
    for (int l = 1; l <= 9; ++l) {
      for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
        const int first = code << (9 - l);
        for (int f = 0; f < (1 << (9 - l)); ++f) look[first + f] = (short)((l << 8) | vals[k]);
      }
      code <<= 1;
    }
    present = true;
    return true;
  }
};

struct JComp { int id, h, v, tq, td, ta, pred, bw, bh, dw, dh; unsigned char* plane; };   // bw x bh: padded plane, dw x dh: real samples

struct JDec {
  const unsigned char* p; const unsigned char* end;
  unsigned bitbuf = 0; int bitcnt = 0; int marker = 0;
  int padbits = 0;          // zero bits at the tail of bitbuf that stand for data the stream does not hold
  bool overrun = false;     // ... and one of them has been consumed: the entropy-coded segment ended early
  unsigned short qt[4][64]; bool qt_ok[4] = {false, false, false, false};
  JHuff dc[4], ac[4];
  JComp comp[3]; int ncomp = 0, W = 0, H = 0, hmax = 1, vmax = 1, restart = 0;
  bool have_sof = false, adobe = false; int adobe_transform = -1;

  void fill() {
    while (bitcnt <= 24) {
      int b = 0;
      if (!marker && p < end) {
        b = *p++;
        if (b == 0xff) {
          int m = p < end ? *p : 0xd9;
          while (m == 0xff && p + 1 < end) { ++p; m = *p; }      // fill bytes
          if (m == 0) ++p;                                        // stuffed zero
          else { marker = m; ++p; b = 0; }
        }
      }
      else padbits += 8;
      bitbuf |= (unsigned)b << (24 - bitcnt);
      bitcnt += 8;
    }
  }
  void used(int n) {
    bitbuf <<= n; bitcnt -= n;
    if (bitcnt < padbits) { overrun = true; padbits = bitcnt < 0 ? 0 : bitcnt; }
  }
  int getbits(int n) {
    if (!n) return 0;
    if (bitcnt < n) fill();
    const int v = (int)(bitbuf >> (32 - n));
    used(n);
    return v;
  }
  int decode(const JHuff& t) {
    if (bitcnt < 16) fill();
    const int lk = t.look[bitbuf >> 23];
    if (lk) { used(lk >> 8); return lk & 0xff; }
    int code = (int)(bitbuf >> 22), l = 10;
    while (l <= 16 && code > t.maxcode[l]) { code = (int)(bitbuf >> (32 - (l + 1))); ++l; }
    if (l > 16) return -1;
    used(l);
    return t.vals[(t.valptr[l] + code - t.mincode[l]) & 255];      // (a corrupt table cannot index past the 256 values)
  }
  static int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }
};

const unsigned char kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline unsigned char clamp8(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
inline int descale(long long x, int n) { return (int)((x + (1LL << (n - 1))) >> n); }

// the accurate integer inverse DCT (13-bit constants, 2 extra bits kept between the passes), output level-shifted by 128
void idct_islow(const int* in, unsigned char* out, int stride) {
  const int CB = 13, P1 = 2;
  const long long F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299, F1_847 = 15137,
                  F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
  int ws[64];
  for (int c = 0; c < 8; ++c) {
    const int* s = in + c;
    int* w = ws + c;
    if (!(s[8] | s[16] | s[24] | s[32] | s[40] | s[48] | s[56])) {
      const int dcv = s[0] * (1 << P1);
      for (int r = 0; r < 8; ++r) w[8 * r] = dcv;
      continue;
    }
    long long z2 = s[16], z3 = s[48];
    long long z1 = (z2 + z3) * F0_541;
    long long tmp2 = z1 - z3 * F1_847, tmp3 = z1 + z2 * F0_765;
    z2 = s[0]; z3 = s[32];
    long long tmp0 = (z2 + z3) * (1 << CB), tmp1 = (z2 - z3) * (1 << CB);
    const long long t10 = tmp0 + tmp3, t13 = tmp0 - tmp3, t11 = tmp1 + tmp2, t12 = tmp1 - tmp2;
    tmp0 = s[56]; tmp1 = s[40]; tmp2 = s[24]; tmp3 = s[8];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    long long z4 = tmp1 + tmp3;
    const long long z5 = (z3 + z4) * F1_175;
    tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
    z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    w[0] = descale(t10 + tmp3, CB - P1); w[56] = descale(t10 - tmp3, CB - P1);
    w[8] = descale(t11 + tmp2, CB - P1); w[48] = descale(t11 - tmp2, CB - P1);
    w[16] = descale(t12 + tmp1, CB - P1); w[40] = descale(t12 - tmp1, CB - P1);
    w[24] = descale(t13 + tmp0, CB - P1); w[32] = descale(t13 - tmp0, CB - P1);
  }
  for (int r = 0; r < 8; ++r) {
    const int* w = ws + 8 * r;
    unsigned char* o = out + (size_t)r * stride;
    long long z2 = w[2], z3 = w[6];
    long long z1 = (z2 + z3) * F0_541;
    long long tmp2 = z1 - z3 * F1_847, tmp3 = z1 + z2 * F0_765;
    long long tmp0 = ((long long)w[0] + w[4]) * (1 << CB), tmp1 = ((long long)w[0] - w[4]) * (1 << CB);
    const long long t10 = tmp0 + tmp3, t13 = tmp0 - tmp3, t11 = tmp1 + tmp2, t12 = tmp1 - tmp2;
    tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    long long z4 = tmp1 + tmp3;
    const long long z5 = (z3 + z4) * F1_175;
    tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
    z1 *= -F0_899; z2 *= -F2_562; z3 *= -F1_961; z4 *= -F0_390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    const int S = CB + P1 + 3;
    o[0] = clamp8(descale(t10 + tmp3, S) + 128); o[7] = clamp8(descale(t10 - tmp3, S) + 128);
    o[1] = clamp8(descale(t11 + tmp2, S) + 128); o[6] = clamp8(descale(t11 - tmp2, S) + 128);
    o[2] = clamp8(descale(t12 + tmp1, S) + 128); o[5] = clamp8(descale(t12 - tmp1, S) + 128);
    o[3] = clamp8(descale(t13 + tmp0, S) + 128); o[4] = clamp8(descale(t13 - tmp0, S) + 128);
  }
}

int jpeg_fail(const char* what) {
  tdg_set_error("tdg_jpeg_decode: %s", what);
  return TDG_EINVAL;
}

// headers up to (and including) the first SOS; returns TDG_OK with d.p behind the SOS header
int jpeg_headers(JDec& d) {
  if (d.end - d.p < 4 || d.p[0] != 0xff || d.p[1] != 0xd8) return jpeg_fail("not a JPEG stream (no SOI marker)");
  d.p += 2;
  for (;;) {
    while (d.p < d.end && *d.p != 0xff) ++d.p;
    while (d.p < d.end && *d.p == 0xff) ++d.p;
    if (d.p >= d.end) return jpeg_fail("no scan (SOS) before the end of the data");
    const int m = *d.p++;
    if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) continue;
    if (m == 0xd9) return jpeg_fail("end of image before any scan");
    if (d.end - d.p < 2) return jpeg_fail("truncated marker segment");
    const int len = (d.p[0] << 8) | d.p[1];
    if (len < 2 || d.end - d.p < len) return jpeg_fail("truncated marker segment");
    const unsigned char* s = d.p + 2;
    const unsigned char* se = d.p + len;
    d.p += len;
    if (m == 0xc0 || m == 0xc1) {
      if (se - s < 6) return jpeg_fail("short frame header");
      if (s[0] != 8) return jpeg_fail("only 8-bit samples are supported");
      d.H = (s[1] << 8) | s[2]; d.W = (s[3] << 8) | s[4]; d.ncomp = s[5];
      if (d.W <= 0 || d.H <= 0) return jpeg_fail("empty image");
      if ((long long)d.W * d.H > (1LL << 26)) return jpeg_fail("images above 64 M pixels are refused (corrupt header?)");
      if (d.ncomp != 1 && d.ncomp != 3) return jpeg_fail("only 1- and 3-component (grayscale, YCbCr / RGB) files are supported");
      if (se - s < 6 + 3 * d.ncomp) return jpeg_fail("short frame header");
      for (int i = 0; i < d.ncomp; ++i) {
        JComp& c = d.comp[i];
        c.id = s[6 + 3 * i]; c.h = s[7 + 3 * i] >> 4; c.v = s[7 + 3 * i] & 15; c.tq = s[8 + 3 * i]; c.plane = nullptr;
        if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) return jpeg_fail("sampling factors other than 1 and 2 are not supported");
        d.hmax = c.h > d.hmax ? c.h : d.hmax; d.vmax = c.v > d.vmax ? c.v : d.vmax;
      }
      // a one-component scan is non-interleaved: one block per MCU in raster order whatever the frame header's
      // sampling factors say (libjpeg decodes such files as if they were 1x1)
      if (d.ncomp == 1) { d.comp[0].h = d.comp[0].v = 1; d.hmax = d.vmax = 1; }
      d.have_sof = true;
    } else if (m == 0xc2) {
      return jpeg_fail("progressive JPEG (SOF2) is not supported");
    } else if (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
      return jpeg_fail("lossless / hierarchical / arithmetic-coded JPEG is not supported");
    } else if (m == 0xcc) {
      return jpeg_fail("arithmetic-coded JPEG is not supported");
    } else if (m == 0xc4) {
      while (s < se) {
        if (se - s < 17) return jpeg_fail("short Huffman table");
        const int tc = s[0] >> 4, th = s[0] & 15;
        if (tc > 1 || th > 3) return jpeg_fail("bad Huffman table id");
        JHuff& t = tc ? d.ac[th] : d.dc[th];
        int n = 0;
        for (int l = 1; l <= 16; ++l) { t.bits[l] = s[l]; n += s[l]; }
        if (n > 256 || se - s < 17 + n) return jpeg_fail("short Huffman table");
        for (int i = 0; i < n; ++i) t.vals[i] = s[17 + i];
        if (!t.build()) return jpeg_fail("bogus Huffman table (its code lengths are not a prefix code)");
        s += 17 + n;
      }
    } else if (m == 0xdb) {
      while (s < se) {
        const int pq = s[0] >> 4, tq = s[0] & 15;
        if (tq > 3 || pq > 1 || se - s < 1 + 64 * (pq + 1)) return jpeg_fail("bad quantisation table");
        for (int i = 0; i < 64; ++i) d.qt[tq][kZigzag[i]] = pq ? (unsigned short)((s[1 + 2 * i] << 8) | s[2 + 2 * i]) : s[1 + i];
        d.qt_ok[tq] = true;
        s += 1 + 64 * (pq + 1);
      }
    } else if (m == 0xdd) {
      if (se - s < 2) return jpeg_fail("short restart interval");
      d.restart = (s[0] << 8) | s[1];
    } else if (m == 0xee) {
      if (se - s >= 12 && s[0] == 'A' && s[1] == 'd' && s[2] == 'o' && s[3] == 'b' && s[4] == 'e') { d.adobe = true; d.adobe_transform = s[11]; }
    } else if (m == 0xda) {
      if (!d.have_sof) return jpeg_fail("scan before the frame header");
      if (se - s < 1 || s[0] != d.ncomp || se - s < 1 + 2 * d.ncomp + 3) return jpeg_fail("a scan that does not hold every component (non-interleaved files are not supported)");
      for (int i = 0; i < d.ncomp; ++i) {
        int k = -1;
        for (int j = 0; j < d.ncomp; ++j) if (d.comp[j].id == s[1 + 2 * i]) k = j;
        if (k != i) return jpeg_fail("scan components out of frame order");
        d.comp[k].td = s[2 + 2 * i] >> 4; d.comp[k].ta = s[2 + 2 * i] & 15;
        if (d.comp[k].td > 3 || d.comp[k].ta > 3 || !d.dc[d.comp[k].td].present || !d.ac[d.comp[k].ta].present || !d.qt_ok[d.comp[k].tq])
          return jpeg_fail("scan refers to a table the file does not define");
      }
      return TDG_OK;
    }
  }
}

// one row of 2:1 horizontal triangle-filter upsampling (n input samples -> 2 n outputs)
void up_h2(const unsigned char* in, int n, unsigned char* out) {
  if (n == 1) { out[0] = out[1] = in[0]; return; }
  out[0] = in[0];
  out[1] = (unsigned char)((in[0] * 3 + in[1] + 2) >> 2);
  for (int i = 1; i < n - 1; ++i) {
    out[2 * i] = (unsigned char)((in[i] * 3 + in[i - 1] + 1) >> 2);
    out[2 * i + 1] = (unsigned char)((in[i] * 3 + in[i + 1] + 2) >> 2);
  }
  out[2 * n - 2] = (unsigned char)((in[n - 1] * 3 + in[n - 2] + 1) >> 2);
  out[2 * n - 1] = in[n - 1];
}

}  // namespace

extern "C" int tdg_jpeg_info(const unsigned char* data, size_t nbytes, int* width, int* height, int* components) {
  if (!data || !width || !height || !components) return jpeg_fail("null argument");
  JDec d;
  d.p = data; d.end = data + nbytes;
  const int st = jpeg_headers(d);
  if (st != TDG_OK) return st;
  *width = d.W; *height = d.H; *components = d.ncomp;
  return TDG_OK;
}

extern "C" int tdg_jpeg_decode(const unsigned char* data, size_t nbytes, unsigned char* rgb, size_t rgb_bytes) {
  if (!data || !rgb) return jpeg_fail("null argument");
  JDec d;
  d.p = data; d.end = data + nbytes;
  int st = jpeg_headers(d);
  if (st != TDG_OK) return st;
  if (rgb_bytes < (size_t)d.W * d.H * 3) return jpeg_fail("output buffer smaller than width * height * 3");
  const int mcuw = 8 * d.hmax, mcuh = 8 * d.vmax;
  const int mx = (d.W + mcuw - 1) / mcuw, my = (d.H + mcuh - 1) / mcuh;
  for (int i = 0; i < d.ncomp; ++i) {
    JComp& c = d.comp[i];
    c.bw = mx * c.h * 8; c.bh = my * c.v * 8;
    c.dw = (d.W * c.h + d.hmax - 1) / d.hmax; c.dh = (d.H * c.v + d.vmax - 1) / d.vmax;
    c.plane = (unsigned char*)malloc((size_t)c.bw * c.bh);
    c.pred = 0;
    if (!c.plane) { for (int j = 0; j < i; ++j) free(d.comp[j].plane); return jpeg_fail("out of memory"); }
  }
  auto cleanup = [&]() { for (int i = 0; i < d.ncomp; ++i) free(d.comp[i].plane); };
  int todo = d.restart, next_rst = 0;
  for (int y = 0; y < my; ++y) {
    for (int x = 0; x < mx; ++x) {
      if (d.restart && todo == 0) {
        d.bitbuf = 0; d.bitcnt = 0; d.padbits = 0;          // byte-align, then the RSTn marker (already seen, or still ahead)
        if (!d.marker) {
          while (d.p + 1 < d.end && !(d.p[0] == 0xff && d.p[1] >= 0xd0 && d.p[1] <= 0xd7)) ++d.p;
          if (d.p + 1 < d.end) { d.marker = d.p[1]; d.p += 2; }
        }
        if (d.marker != 0xd0 + next_rst) { cleanup(); return jpeg_fail("missing or out-of-order restart marker"); }
        d.marker = 0; next_rst = (next_rst + 1) & 7; todo = d.restart;
        for (int i = 0; i < d.ncomp; ++i) d.comp[i].pred = 0;
      }
      for (int i = 0; i < d.ncomp; ++i) {
        JComp& c = d.comp[i];
        for (int by = 0; by < c.v; ++by)
          for (int bx = 0; bx < c.h; ++bx) {
            int blk[64] = {0};
            const int t = d.decode(d.dc[c.td]);
            if (t < 0 || t > 11) { cleanup(); return jpeg_fail("corrupt entropy-coded data (DC)"); }
            c.pred = (int)((unsigned)c.pred + (unsigned)(t ? JDec::extend(d.getbits(t), t) : 0));     // (a corrupt stream may wrap; never UB)
            blk[0] = (int)((long long)c.pred * d.qt[c.tq][0]);
            for (int k = 1; k < 64;) {
              const int rs = d.decode(d.ac[c.ta]);
              if (rs < 0) { cleanup(); return jpeg_fail("corrupt entropy-coded data (AC)"); }
              const int r = rs >> 4, s = rs & 15;
              if (!s) { if (r != 15) break; k += 16; continue; }
              k += r;
              if (k > 63) { cleanup(); return jpeg_fail("corrupt entropy-coded data (run past the block)"); }
              const int z = kZigzag[k];
              blk[z] = JDec::extend(d.getbits(s), s) * d.qt[c.tq][z];
              ++k;
            }
            // tf.image.decode_image raises on a file cut inside its scan; so does this decoder (no grey tail)
            if (d.overrun) { cleanup(); return jpeg_fail("the entropy-coded data ends (or meets a marker) before the last block"); }
            idct_islow(blk, c.plane + (size_t)((y * c.v + by) * 8) * c.bw + (x * c.h + bx) * 8, c.bw);
          }
      }
      if (d.restart) --todo;
    }
  }
  // upsample every component to full resolution (triangle filter for 2:1, as libjpeg's default), then colour-convert
  unsigned char* full[3] = {nullptr, nullptr, nullptr};
  bool own[3] = {false, false, false};
  for (int i = 0; i < d.ncomp; ++i) {
    JComp& c = d.comp[i];
    const int fh = d.hmax / c.h, fv = d.vmax / c.v;
    if (fh == 1 && fv == 1) { full[i] = c.plane; continue; }
    const int ow = fh * c.dw, oh = fv * c.dh;               // >= W, H
    unsigned char* o = (unsigned char*)malloc((size_t)ow * oh);
    if (!o) { for (int j = 0; j < i; ++j) if (own[j]) free(full[j]); cleanup(); return jpeg_fail("out of memory"); }
    for (int r = 0; r < c.dh; ++r) {
      const unsigned char* in0 = c.plane + (size_t)r * c.bw;
      if (fv == 1) {                                          // h2v1
        up_h2(in0, c.dw, o + (size_t)r * ow);
        continue;
      }
      for (int v = 0; v < 2; ++v) {
        int rn = v == 0 ? r - 1 : r + 1;                      // the nearer neighbour row (edges replicate)
        rn = rn < 0 ? 0 : (rn > c.dh - 1 ? c.dh - 1 : rn);
        const unsigned char* in1 = c.plane + (size_t)rn * c.bw;
        unsigned char* out = o + (size_t)(2 * r + v) * ow;
        if (fh == 1) {                                        // h1v2
          const int bias = v == 0 ? 1 : 2;
          for (int i2 = 0; i2 < c.dw; ++i2) out[i2] = (unsigned char)((in0[i2] * 3 + in1[i2] + bias) >> 2);
          continue;
        }
        const int n = c.dw;                                   // h2v2
        if (n == 1) { const int t = in0[0] * 3 + in1[0]; out[0] = (unsigned char)((t * 4 + 8) >> 4); out[1] = (unsigned char)((t * 4 + 7) >> 4); continue; }
        int last = in0[0] * 3 + in1[0], cur = in0[1] * 3 + in1[1];
        out[0] = (unsigned char)((last * 4 + 8) >> 4);
        out[1] = (unsigned char)((last * 3 + cur + 7) >> 4);
        for (int i2 = 1; i2 < n - 1; ++i2) {
          const int nxt = in0[i2 + 1] * 3 + in1[i2 + 1];
          out[2 * i2] = (unsigned char)((cur * 3 + last + 8) >> 4);
          out[2 * i2 + 1] = (unsigned char)((cur * 3 + nxt + 7) >> 4);
          last = cur; cur = nxt;
        }
        out[2 * n - 2] = (unsigned char)((cur * 3 + last + 8) >> 4);
        out[2 * n - 1] = (unsigned char)((cur * 4 + 7) >> 4);
      }
    }
    full[i] = o; own[i] = true;
  }
  const bool ycc = d.ncomp == 3 && !(d.adobe && d.adobe_transform == 0) &&
                   !(d.comp[0].id == 'R' && d.comp[1].id == 'G' && d.comp[2].id == 'B' && !d.adobe);
  for (int yy = 0; yy < d.H; ++yy) {
    unsigned char* o = rgb + (size_t)yy * d.W * 3;
    if (d.ncomp == 1) {
      const unsigned char* g = full[0] + (size_t)yy * d.comp[0].bw;
      for (int xx = 0; xx < d.W; ++xx) { o[3 * xx] = o[3 * xx + 1] = o[3 * xx + 2] = g[xx]; }
      continue;
    }
    const unsigned char* c0 = full[0] + (size_t)yy * (own[0] ? (d.hmax / d.comp[0].h) * d.comp[0].dw : d.comp[0].bw);
    const unsigned char* c1 = full[1] + (size_t)yy * (own[1] ? (d.hmax / d.comp[1].h) * d.comp[1].dw : d.comp[1].bw);
    const unsigned char* c2 = full[2] + (size_t)yy * (own[2] ? (d.hmax / d.comp[2].h) * d.comp[2].dw : d.comp[2].bw);
    for (int xx = 0; xx < d.W; ++xx) {
      if (!ycc) { o[3 * xx] = c0[xx]; o[3 * xx + 1] = c1[xx]; o[3 * xx + 2] = c2[xx]; continue; }
      const int Y = c0[xx], cb = c1[xx] - 128, cr = c2[xx] - 128;
      const int r = Y + ((91881 * cr + 32768) >> 16);
      const int g = Y + ((-22554 * cb - 46802 * cr + 32768) >> 16);
      const int b = Y + ((116130 * cb + 32768) >> 16);
      o[3 * xx] = clamp8(r); o[3 * xx + 1] = clamp8(g); o[3 * xx + 2] = clamp8(b);
    }
  }
  for (int i = 0; i < d.ncomp; ++i) if (own[i]) free(full[i]);
  cleanup();
  return TDG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Shuffle buffer over example indices + batch assembly (include/tdg.h: tdg_shuffle_draw / tdg_gather_rows).
namespace {
inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
inline uint64_t xoshiro_next(uint64_t* s) {
  const uint64_t r = rotl64(s[1] * 5, 7) * 9, t = s[1] << 17;
  s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
  s[2] ^= t;
  s[3] = rotl64(s[3], 45);
  return r;
}
// uniform in [0, n): Lemire's multiply-shift with rejection (unbiased)
inline uint64_t uniform_below(uint64_t* s, uint64_t n) {
  uint64_t x = xoshiro_next(s);
  __uint128_t m = (__uint128_t)x * n;
  uint64_t lo = (uint64_t)m;
  if (lo < n) {
    const uint64_t t = (0 - n) % n;
    while (lo < t) {
      x = xoshiro_next(s);
      m = (__uint128_t)x * n;
      lo = (uint64_t)m;
    }
  }
  return (uint64_t)(m >> 64);
}
}  // namespace

extern "C" int tdg_shuffle_draw(int64_t* buf, int64_t buf_len, uint64_t* state, int64_t* next_in, int64_t n_total, int64_t count,
                                int64_t* out) {
  if (!buf || !state || !next_in || !out || buf_len < 1 || n_total < 1 || count < 0 || *next_in < 0 || *next_in >= n_total) {
    tdg_set_error("tdg_shuffle_draw: bad argument (buf_len %lld, n_total %lld, count %lld)", (long long)buf_len, (long long)n_total, (long long)count);
    return TDG_EINVAL;
  }
  if (!(state[0] | state[1] | state[2] | state[3])) { tdg_set_error("tdg_shuffle_draw: the all-zero generator state is invalid"); return TDG_EINVAL; }
  int64_t nx = *next_in;
  for (int64_t i = 0; i < count; ++i) {
    const int64_t j = (int64_t)uniform_below(state, (uint64_t)buf_len);
    out[i] = buf[j];
    buf[j] = nx;
    nx = nx + 1 == n_total ? 0 : nx + 1;
  }
  *next_in = nx;
  return TDG_OK;
}

extern "C" int tdg_gather_rows(const unsigned char* src, int64_t n_rows, size_t row_bytes, const int64_t* idx, int64_t count,
                               unsigned char* out) {
  if (!src || !idx || !out || n_rows < 1 || count < 0) { tdg_set_error("tdg_gather_rows: bad argument"); return TDG_EINVAL; }
  for (int64_t i = 0; i < count; ++i) {
    if (idx[i] < 0 || idx[i] >= n_rows) { tdg_set_error("tdg_gather_rows: index %lld outside [0, %lld)", (long long)idx[i], (long long)n_rows); return TDG_EINVAL; }
    memcpy(out + (size_t)i * row_bytes, src + (size_t)idx[i] * row_bytes, row_bytes);
  }
  return TDG_OK;
}
