// Implicit-GEMM kernel argument blocks shared by tdg_igemm.hip (device) and tdg_conv.cpp-side
// planning code (host).  gfx950 only.
#pragma once
#include "tdg_common.h"

#define IG_MAX_TAPS 32
#define IG_MAX_CLASSES 4
#define IG_BKB 128   // bytes of K per LDS row per main-loop step (8 x 16-byte chunks)

// One output-parity class of a GEMM  C[M][N] = gather(A)[M][K] * Bp[N][K]^T.
//   rows  m  <-> anchor (nb, a, b) on a [GH, GW] grid per image
//   k        <-> (tap t, channel c): source pixel (a*sigma + dh[t], b*sigma + dw[t]), zero outside
//   output pixel of row m: (a*os + oh0, b*os + ow0)
struct IgClass {
  int M, GH, GW, ntaps;
  int K;                 // ntaps * C (logical K, elements)
  int nsteps;            // ceil(K / (IG_BKB / sizeof(T)))
  int Kp;                // packed filter row pitch (elements) = nsteps * BKE
  int oh0, ow0;
  unsigned w_off_bytes;  // this class's packed filter block inside the packed buffer
  FastDiv fd_ghw, fd_gw, fd_nw;
  FastDiv fd_nt;         // divide by ntaps (K order of a sliced filter: see IgArgs.fd_ck)
  // taps form a grid: tap t = th*nw + tw reads source offset (dh0 + sh*th, dw0 + sw*tw)
  int nh, nw, dh0, dw0, sh, sw;
  short tap[IG_MAX_TAPS];  // (dh & 0xff) | ((dw & 0xff) << 8), signed bytes
  // igemm_fwd_patch_kernel: the taps fall into <= 4 groups of consecutive taps that read one sub-lattice (ph, pw) of the
  // source (stride-2 forward: the four pixel parities; stride-1 gathers: one group); tap t of group g reads lattice pixel
  // (a + dhq, b + dwq) of a [QH, QW] lattice per image, dhq = (dh - ph) / sigma
  int ngroups, QH, QW;
  FastDiv fd_qhw, fd_qw;   // divide by QH * QW, by QW
  // block tiles of igemm_fwd_patch_kernel (nbh * nbw > 1; 0 / 1: whole images): GH, GW, fd_ghw, fd_gw then describe ONE BLOCK of
  // an image's anchor grid, rows m run block by block (nbh x nbw blocks per image), and [QH, QW] is the block's lattice
  // window incl. `halo` pixels on every side
  int nbh, nbw, halo;
  struct { int t0, nt, ph, pw; } grp[4];   // (ints: the kernel reads them with scalar dword loads)
};

struct IgArgs {
  const void* src;
  const void* wpack;
  const float* bias;
  void* out;
  const void* mask_src;
  unsigned src_bytes, w_bytes;
  int SH, SW, sigma;
  int C, Cs;             // channels per tap in K (effective, multiple of VEC on the vector path), channel stride
  FastDiv fd_c;          // divide by C (scalar path) or by C/VEC (vector path)
  // K order of the packed filter on the vector path: K chunk (16 bytes) index = (slice * ntaps + tap) * ckv + j with
  // ckv = fd_ck.d chunks of a tap's channels per slice, channel vector cv = slice * ckv + j.  nslices = 1 (ckv = C/VEC)
  // is the plain (tap, channel) order; a sliced filter (tdg_k_slice_chunks) lets igemm_fwd_patch_kernel keep a
  // channel slice of the gathered operand resident in LDS across all the taps that read it.
  FastDiv fd_ck;
  int nslices;
  int N, OH, OW, os, Cso;
  int act, mask_mode, accumulate;
  float leak;
  int ntiles_n, ntiles_m_max, nclasses;
  int n_begin;           // first output column of this launch (a problem's columns may be split over launches)
  // split-K (small-M problems that would leave most CUs idle): blockIdx.y = K split, f32 partial tiles go to
  // slab[(split * nclasses + class) * slab_rows + m][N]; splitk_finish_kernel sums them and applies the epilogue
  int ksplit, steps_per_split, slab_rows;
  float* slab;
  float* col_partial;    // per row tile [2][N] column sums of the stored tile (TdgEpilogue.col_partial), or null
  int col_mode;          // TDG_COL_*
  int col_images;        // rows of images >= col_images do not count (0: all)
  unsigned long long* stamps;  // diagnostic build (-DTDG_STAMPS) only: per-wave cycle sums; null otherwise
  int debug;             // TDG_DEBUG_ABLATE (diagnostics only): 1 no global loads in loop, 2 + no LDS stores, 3 no MFMA
  IgClass cls[IG_MAX_CLASSES];
};

// Filter-gradient GEMM  out[(t,c)][n] = sum_m gather(A)[m][(t,c)] * G[m][n], split over m.
struct WgArgs {
  const void* src;       // gathered operand (big-side tensor)
  const void* src2;      // second gathered tensor: rows m >= m_switch are image (m / GHW - img_switch) of it (or null)
  const void* g;         // dense rows [M][Gs] (small-side tensor)
  float* slabs;          // [nsplit][ntaps*Clog][N] f32 partials
  unsigned src_bytes, src2_bytes, g_bytes;
  int m_switch, img_switch;   // (m_switch = M and src2 = null: one source)
  int M, GH, GW, SH, SW, sigma;
  int C, Clog, Cs, ntaps;  // effective / logical channels per tap, channel stride
  int KK;                  // ntaps * C
  FastDiv fd_c, fd_ghw, fd_gw;
  int N, Gs;               // G channels (effective), row pitch
  int Nlog;                // logical N written to the slabs
  int nsplit, m_per_split; // m_per_split is a multiple of the step's row count
  long long slab_stride;   // elements
  int ntiles_n, ntiles_k;
  unsigned long long* stamps;   // diagnostic build (-DTDG_STAMPS) only; null otherwise
  short tap[IG_MAX_TAPS];
};
