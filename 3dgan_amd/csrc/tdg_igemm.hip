// Implicit-GEMM convolution kernels for gfx950 (CDNA4): conv2d forward, conv2d_backprop_input
// (= conv2d_transpose) and conv2d_backprop_filter of the reference's conv stack
// (ops/layers.py:101,142 and the autodiff ops TF derives from them).
//
// Design:
//  * 64-wide waves, 4 waves per workgroup, v_mfma_f32_16x16x32_bf16 (bf16) or
//    v_mfma_f32_16x16x4_f32 (exact f32 parity path); f32 accumulators.
//  * NHWC activations: the im2col gather is a 16-byte vector load per (pixel, 8 channels)
//    through a buffer descriptor, so SAME-padding taps are out-of-range offsets that read 0.
//  * Operands are staged through LDS in 128-byte K rows with an XOR swizzle that makes every
//    ds_read_b128 fragment read conflict-free (chunk ^= (row >> 1) & 7).
//  * MFMA operands are swapped (filter rows first) so each lane ends up owning 4 consecutive
//    output channels of one pixel: the epilogue (bias, activation, derivative mask) stores
//    8/16-byte vectors.
//  * Backward-data runs as `stride^2` output-parity classes of stride-1 gathers (blockIdx.z),
//    each with its own tap subset and packed filter block: no zero-stuffed MACs.
//  * Filter gradient: both operands are transposed on the way out of LDS
//    (ds_read_b64_tr_b16 for bf16, ds_read_b32 for f32), rows split over blockIdx.z into f32
//    slabs that a second kernel sums in a fixed order (deterministic, no atomics).
#include <stdlib.h>

#include "tdg_igemm.h"
#include "tdg_wgrad_patch.h"
#include <type_traits>

#define OOB_OFFSET 0xFFFFFF00u

// In-kernel cycle stamps: compiled only into the diagnostic library (build.sh stamps); no stamp executes in the product build.
#ifdef TDG_STAMPS
#define TDG_STAMP(var)                                                                            \
  do {                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");                  \
    __builtin_amdgcn_sched_barrier(0);                                                            \
  } while (0)
#else
#define TDG_STAMP(var) do { } while (0)
#endif

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(IntC<I>{});
    static_for<I + 1, N>(f);
  }
}

template <typename T>
struct Mma;
template <>
struct Mma<bf16_t> {
  using frag = bf16x8;
  static __device__ __forceinline__ void run(f32x4& acc, const frag& a, const frag& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  }
};
template <>
struct Mma<float> {
  using frag = f32x4;
  static __device__ __forceinline__ void run(f32x4& acc, const frag& a, const frag& b) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
  }
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// Raw buffer descriptor as four SGPR dwords, and an LDS-DMA load (`buffer_load_dwordx4 ... offen lds`) issued
// from inline asm.  hipcc tracks the builtin form as a pending LDS write and puts `s_waitcnt vmcnt(0)` in front
// of the next `ds_read_b64_tr_b16` (the transposing reads of the filter-gradient kernel), which serialises load and
// multiply; the asm form is invisible to that pass, so the kernel drains it by hand (vmcnt before its barrier).
__device__ __forceinline__ i32x4 make_rsrc_words(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void lds_dma_b128(const i32x4& rsrc, unsigned voff, unsigned lds_byte) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_byte) : "memory");
}

template <typename T>
__device__ __forceinline__ T buffer_load_elem(__amdgpu_buffer_rsrc_t r, unsigned off);
template <>
__device__ __forceinline__ float buffer_load_elem<float>(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
template <>
__device__ __forceinline__ bf16_t buffer_load_elem<bf16_t>(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(bf16_t, __builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0));
}

__device__ __forceinline__ int tap_dh(int pk) { return (pk << 24) >> 24; }
__device__ __forceinline__ int tap_dw(int pk) { return (pk << 16) >> 24; }

// swizzled byte address of 16-byte chunk `chunk` of row `row` in a [rows][128 B] LDS tile
__device__ __forceinline__ int lds_swz(int row, int chunk) {
  return row * IG_BKB + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// K chunk (16 bytes of a packed filter row) -> (tap index in the class's tap table, channel vector of the tap), for the
// plain (tap, channel) order and the sliced order of IgArgs.fd_ck alike: chunk = (slice * ntaps + tap) * ckv + j.
struct KChunk { int tap, cv; bool ok; };
__device__ __forceinline__ KChunk k_decode(unsigned chunk, const FastDiv& fd_ck, const FastDiv& fd_nt, int nslices) {
  const unsigned u = fd_div(chunk, fd_ck);
  const unsigned j = chunk - u * fd_ck.d;
  const unsigned sl = fd_div(u, fd_nt);
  KChunk r;
  r.tap = (int)(u - sl * fd_nt.d);
  r.cv = (int)(sl * fd_ck.d + j);
  r.ok = (int)sl < nslices;
  return r;
}

// ============================================================================================
// forward-type implicit GEMM: conv2d fwd, conv2d bwd-data (parity classes), dense
// ============================================================================================
template <typename T, int BM, int BN, int WGM, int WGN, bool VECA>
__global__ void __launch_bounds__(256, 2) igemm_fwd_kernel(const IgArgs args) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int BKE = IG_BKB / (int)sizeof(T);
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int NA = BM / 32;              // A vectors per thread per step (vector path)
  constexpr int NB = (BN + 31) / 32;       // B vectors per thread per step
  static_assert(WGM * WGN == 4 && WM % 16 == 0 && WN % 16 == 0 && BM % 32 == 0, "tile config");
  using Frag = typename Mma<T>::frag;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sB = smem + BM * IG_BKB;
  int* sTap = reinterpret_cast<int*>(smem + (BM + BN) * IG_BKB);

  const IgClass& cl = args.cls[blockIdx.z];
  const int tid = threadIdx.x;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = bid / args.ntiles_n;
  const int tile_n = bid - tile_m * args.ntiles_n;
  const int M = cl.M;
  if (tile_m * BM >= M) return;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int ntaps = cl.ntaps;
  const int SH = args.SH, SW = args.SW, Cs = args.Cs;

  if (tid < IG_MAX_TAPS) sTap[tid] = cl.tap[tid];
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rA = make_rsrc(args.src, args.src_bytes);
  const __amdgpu_buffer_rsrc_t rB =
      make_rsrc(static_cast<const char*>(args.wpack) + cl.w_off_bytes, args.w_bytes - cl.w_off_bytes);

  // ---- per-thread row bookkeeping (vector path: chunk column fixed, NA rows) ----------------
  const int ch = tid & 7;
  const int rsub = tid >> 3;
  int a_h[NA], a_w[NA];
  unsigned a_base[NA];
  if constexpr (VECA) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int m = m0 + rsub + 32 * i;
      const bool ok = m < M;
      const unsigned mm = ok ? (unsigned)m : 0u;
      const unsigned nb = fd_div(mm, cl.fd_ghw);
      const unsigned rem = mm - nb * (unsigned)(cl.GH * cl.GW);
      const unsigned a = fd_div(rem, cl.fd_gw);
      const unsigned b = rem - a * (unsigned)cl.GW;
      a_h[i] = ok ? (int)a * args.sigma : -(1 << 20);
      a_w[i] = (int)b * args.sigma;
      a_base[i] = ((nb * (unsigned)SH + a * (unsigned)args.sigma) * (unsigned)SW + b * (unsigned)args.sigma) * (unsigned)Cs;
    }
  }
  // k decomposition of this thread's chunk column: kv = step*8 + ch -> (tap, cvec)
  const int CV = (int)args.fd_c.d;  // vectors per tap (vector path) / channels per tap (scalar path)

  i32x4 ra[NA], rb[NB];

  auto load_tiles = [&](int step) {
    if constexpr (VECA) {
      const KChunk kc = k_decode((unsigned)(step * 8 + ch), args.fd_ck, cl.fd_nt, args.nslices);
      const bool t_ok = kc.ok;
      const int pk = sTap[t_ok ? kc.tap : 0];
      const int dh = tap_dh(pk), dw = tap_dw(pk);
      const int koff = (dh * SW + dw) * Cs + kc.cv * VEC;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int ih = a_h[i] + dh, iw = a_w[i] + dw;
        const bool ok = t_ok && (unsigned)ih < (unsigned)SH && (unsigned)iw < (unsigned)SW;
        const unsigned off = ok ? (a_base[i] + (unsigned)koff) * (unsigned)sizeof(T) : OOB_OFFSET;
        ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rA, off, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int rbn = rsub + 32 * j;
      const int n = n0 + rbn;
      const bool ok = (rbn < BN) && (n < args.N);
      const unsigned off =
          ok ? ((unsigned)n * (unsigned)cl.Kp + (unsigned)(step * BKE + ch * VEC)) * (unsigned)sizeof(T) : OOB_OFFSET;
      rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rB, off, 0, 0);
    }
  };

  auto store_tiles = [&]() {
    if constexpr (VECA) {
#pragma unroll
      for (int i = 0; i < NA; ++i) *reinterpret_cast<i32x4*>(sA + lds_swz(rsub + 32 * i, ch)) = ra[i];
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int rbn = rsub + 32 * j;
      if (rbn < BN) *reinterpret_cast<i32x4*>(sB + lds_swz(rbn, ch)) = rb[j];
    }
  };

  // scalar gather for thin-channel sources (C in {1,3,4,...}): element-by-element, no prefetch
  auto gather_scalar = [&](int step) {
    constexpr int RPT = BM * BKE / 256;   // elements per thread
    constexpr int RSTEP = 256 / BKE;      // row stride between a thread's elements
    const int kcol = tid % BKE;
    const int k = step * BKE + kcol;
    const bool k_ok = k < cl.K;
    const unsigned tap = fd_div((unsigned)(k_ok ? k : 0), args.fd_c);
    const int c = (k_ok ? k : 0) - (int)tap * CV;
    const int pk = sTap[tap];
    const int dh = tap_dh(pk), dw = tap_dw(pk);
    const int bcol = kcol * (int)sizeof(T);
#pragma unroll 4
    for (int i = 0; i < RPT; ++i) {
      const int row = tid / BKE + RSTEP * i;
      const int m = m0 + row;
      const bool okm = m < M;
      const unsigned mm = okm ? (unsigned)m : 0u;
      const unsigned nb = fd_div(mm, cl.fd_ghw);
      const unsigned rem = mm - nb * (unsigned)(cl.GH * cl.GW);
      const unsigned a = fd_div(rem, cl.fd_gw);
      const unsigned b = rem - a * (unsigned)cl.GW;
      const int ih = (int)a * args.sigma + dh, iw = (int)b * args.sigma + dw;
      const bool ok = okm && k_ok && (unsigned)ih < (unsigned)SH && (unsigned)iw < (unsigned)SW;
      const unsigned off =
          ok ? (((nb * (unsigned)SH + (unsigned)ih) * (unsigned)SW + (unsigned)iw) * (unsigned)Cs + (unsigned)c) *
                   (unsigned)sizeof(T)
             : OOB_OFFSET;
      const T v = buffer_load_elem<T>(rA, off);
      *reinterpret_cast<T*>(sA + lds_swz(row, bcol >> 4) + (bcol & 15)) = v;
    }
  };

  // ---- wave / lane roles ------------------------------------------------------------------------
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WGN, wn = wave - wm * WGN;
  const int r16 = lane & 15, q = lane >> 4;
  const int swl = (r16 >> 1) & 7;                       // swizzle term (tile bases are multiples of 16 rows)
  const char* pA = sA + (wm * WM + r16) * IG_BKB;
  const char* pB = sB + (wn * WN + r16) * IG_BKB;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nsteps = cl.nsteps;
  load_tiles(0);
  const int dbg = args.debug;
  for (int step = 0; step < nsteps; ++step) {
    if (dbg < 2 || step == 0) store_tiles();
    if constexpr (!VECA) gather_scalar(step);
    __syncthreads();
    if (step + 1 < nsteps && (dbg == 0 || dbg == 3)) load_tiles(step + 1);
    if (dbg != 3)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int coff = ((ks * 4 + q) ^ swl) << 4;
      Frag fa[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const Frag*>(pA + i * 16 * IG_BKB + coff);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const Frag fb = *reinterpret_cast<const Frag*>(pB + j * 16 * IG_BKB + coff);
#pragma unroll
        for (int i = 0; i < TM; ++i) Mma<T>::run(acc[i][j], fb, fa[i]);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: lane owns pixel (r16) x 4 consecutive channels (q*4..) per 16x16 tile ----------
  const int N = args.N, Cso = args.Cso;
  const bool vec_ok = ((N & 3) == 0) && ((Cso & 3) == 0);
  T* out = static_cast<T*>(args.out);
  const T* msk = static_cast<const T*>(args.mask_src);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WM + i * 16 + r16;
    if (m >= M) continue;
    const unsigned nb = fd_div((unsigned)m, cl.fd_ghw);
    const unsigned rem = (unsigned)m - nb * (unsigned)(cl.GH * cl.GW);
    const unsigned a = fd_div(rem, cl.fd_gw);
    const unsigned b = rem - a * (unsigned)cl.GW;
    const size_t pix = ((size_t)(nb * (unsigned)args.OH + a * (unsigned)args.os + (unsigned)cl.oh0) * (unsigned)args.OW +
                        b * (unsigned)args.os + (unsigned)cl.ow0) * (size_t)Cso;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WN + j * 16 + q * 4;
      if (n >= N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (vec_ok) {
        if (args.bias) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(args.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bv[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], args.act, args.leak);
        if (args.accumulate) {
          if constexpr (sizeof(T) == 4) {
            const f32x4 ov = *reinterpret_cast<const f32x4*>(out + pix + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += ov[e];
          } else {
            const bf16x4 ov = *reinterpret_cast<const bf16x4*>(out + pix + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)ov[e];
          }
        }
        if (args.mask_mode != TDG_MASK_NONE) {
          if constexpr (sizeof(T) == 4) {
            const f32x4 mv = *reinterpret_cast<const f32x4*>(msk + pix + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= mask_factor(mv[e], args.mask_mode, args.leak);
          } else {
            const bf16x4 mv = *reinterpret_cast<const bf16x4*>(msk + pix + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= mask_factor((float)mv[e], args.mask_mode, args.leak);
          }
        }
        if constexpr (sizeof(T) == 4) {
          *reinterpret_cast<f32x4*>(out + pix + n) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
          *reinterpret_cast<bf16x4*>(out + pix + n) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e < N) {
            float x = v[e] + (args.bias ? args.bias[n + e] : 0.f);
            x = apply_act(x, args.act, args.leak);
            if (args.accumulate) x += to_f32<T>(out[pix + n + e]);
            if (args.mask_mode != TDG_MASK_NONE) x *= mask_factor(to_f32<T>(msk[pix + n + e]), args.mask_mode, args.leak);
            out[pix + n + e] = from_f32<T>(x);
          }
        }
      }
    }
  }
}

// ============================================================================================
// Large-M variant of the forward-type GEMM: 256 x BN tile, 8 waves (one 32-row strip each),
// operands streamed by LDS-DMA (`buffer_load_dwordx4 ... lds`, no VGPR staging, no ds_write) into a
// 2-stage LDS ring, ONE barrier per K step.  LDS-DMA writes lane-linear (base + lane*16), so the
// bank-conflict swizzle is applied to each lane's SOURCE chunk: a wave instruction fills 8 LDS rows
// x 8 physical chunks, lane l supplies logical chunk (l&7) ^ ((row>>1)&7).  Instructions are dealt
// to waves by parity of their index so that this XOR term is one constant per lane.
// ============================================================================================
// Per-lane state of the LDS-DMA loader of igemm_fwd_dma_kernel.  A K step's loads are NP single
// wave-instructions ("pieces": NAJ of the gathered operand, NBJ of the packed filter; every wave issues
// the same number, a wave with one filter piece less issues an out-of-range one into don't-care rows), so
// that the K loop can place each piece between two MFMA groups instead of stalling on all of them at
// once: with the pieces in one block the stamps showed 35-43 % of a step spent issuing them (the
// texture path takes a 1 KiB piece every ~150 cycles with eight waves queueing) and no MFMA running.
template <typename T, int BM, int BN, int NW = 8>
struct DmaLoader {
  static constexpr int NAJ = BM / (8 * NW);                 // 8-row pieces of the gathered operand per wave
  static constexpr int BNL = 2 * ((BN / 16 + 1) / 2) * 16;
  static constexpr int NIB = BNL / 8;
  static constexpr int NBJ = (NIB + NW - 1) / NW;
  static constexpr int NP = NAJ + NBJ;
  static constexpr int VEC = 16 / (int)sizeof(T);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  __amdgpu_buffer_rsrc_t rA, rB;
  int a_h[NAJ], a_w[NAJ];
  unsigned a_base[NAJ];
  unsigned b_row[NBJ];          // byte offset of this lane's chunk at step 0; OOB_OFFSET: row outside N or dummy piece
  int a_lds[NAJ], b_lds[NBJ];   // wave-uniform byte offsets inside a stage
  int SH, SW, Cs, CV, ntaps, lch;
  int tapreg;                   // the class's tap table, entry (lane & 31) in each lane
  FastDiv fd_ck, fd_nt;         // K order of the packed filter (IgArgs.fd_ck)
  int nslices;
  // state of the step being loaded
  int dh, dw, t_ok;
  unsigned koff, kbyte;
  int step0;                    // first K step of this workgroup's split (0 without split-K)

  // K chunk (step, lch) -> tap (th, tw) and 16-byte vector cv inside the tap's channels
  // (the tap's source offset comes from the class's tap table: the table's ORDER is the K order of the packed
  //  filter, which the host is free to choose -- see fwd_tap_order)
  __device__ __forceinline__ void prepare(int step) {
    step += step0;
    const KChunk kc = k_decode((unsigned)(step * 8 + lch), fd_ck, fd_nt, nslices);
    const int cv = kc.cv;
    t_ok = kc.ok;
    const int pk = __builtin_amdgcn_ds_bpermute((t_ok ? kc.tap : 0) << 2, tapreg);    // lane i holds tap i: a lane crossbar
    dh = tap_dh(pk);                                                                  // read, not an LDS access the
    dw = tap_dw(pk);                                                                  // compiler would fence the DMA for
    koff = (unsigned)((dh * SW + dw) * Cs + cv * VEC);                                // (requesting the entry a step ahead
    kbyte = (unsigned)step * IG_BKB;                                                  //  was measured: slower)
  }
  template <int P>
  __device__ __forceinline__ void piece(char* stage) const {
    if constexpr (P < NAJ) {
      const int ih = a_h[P] + dh, iw = a_w[P] + dw;
      const int ok = t_ok & ((unsigned)ih < (unsigned)SH) & ((unsigned)iw < (unsigned)SW);
      const unsigned off = ok ? (a_base[P] + koff) * (unsigned)sizeof(T) : OOB_OFFSET;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(stage + a_lds[P]), 16, off, 0, 0, 0);
    } else if constexpr (P < NP) {
      constexpr int j = P - NAJ;
      const unsigned off = b_row[j] == OOB_OFFSET ? OOB_OFFSET : b_row[j] + kbyte;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr_t)(stage + b_lds[j]), 16, off, 0, 0, 0);
    }
  }
  template <int P = 0>
  __device__ __forceinline__ void all_pieces(char* stage) const {
    if constexpr (P < NP) {
      piece<P>(stage);
      all_pieces<P + 1>(stage);
    }
  }
  template <int P = NAJ>
  __device__ __forceinline__ void b_pieces(char* stage) const {      // diagnostics (TDG_DEBUG_ABLATE=7): the filter pieces only
    if constexpr (P < NP) {
      piece<P>(stage);
      b_pieces<P + 1>(stage);
    }
  }
};

// one K step (2 x 64-byte halves) of a wave's TM x TN tiles, software-pipelined by hand: the B
// fragment of tile t+2 (and the A fragments of the second half) are requested before the MFMAs of
// tile t issue, and sched_group_barrier pins that interleave, so an MFMA group never waits for a
// read issued right in front of it.  With LOADS, loader piece t / ILV is issued behind the MFMAs of tile t.
template <typename T, int TM, int TN, int t, bool LOADS, int ILV, typename LD>
__device__ __forceinline__ void dma_mma_tile(f32x4 (&acc)[TM][TN], typename Mma<T>::frag (&fa)[2][TM],
                                             typename Mma<T>::frag (&fb)[2 * TN], const char* pA, const char* pB, int coff0,
                                             int coff1, const LD& ld, char* nstage) {
  using Frag = typename Mma<T>::frag;
  constexpr int NT = 2 * TN;
  if constexpr (t < NT) {
    constexpr int ks = t / TN, j = t - ks * TN;
    if constexpr (t + 2 < NT) {
      constexpr int ks2 = (t + 2) / TN, j2 = (t + 2) - ks2 * TN;
      fb[t + 2] = *reinterpret_cast<const Frag*>(pB + j2 * 16 * IG_BKB + (ks2 ? coff1 : coff0));
    }
    if constexpr (t < TM) fa[1][t] = *reinterpret_cast<const Frag*>(pA + t * 16 * IG_BKB + coff1);
#pragma unroll
    for (int i = 0; i < TM; ++i) Mma<T>::run(acc[i][j], fb[t], fa[ks][i]);
    constexpr bool has_piece = LOADS && (t % ILV == 0) && (t / ILV < LD::NP);
    if constexpr (has_piece) ld.template piece<t / ILV>(nstage);
    __builtin_amdgcn_sched_group_barrier(0x100, (t + 2 < NT ? 1 : 0) + (t < TM ? 1 : 0), 0);
    __builtin_amdgcn_sched_group_barrier(0x008, TM * (sizeof(T) == 4 ? 4 : 1), 0);
    if constexpr (has_piece) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    dma_mma_tile<T, TM, TN, t + 1, LOADS, ILV, LD>(acc, fa, fb, pA, pB, coff0, coff1, ld, nstage);
  }
}

template <typename T, int TM, int TN, bool LOADS, int ILV, typename LD>
__device__ __forceinline__ void dma_mma_step(f32x4 (&acc)[TM][TN], const char* pA, const char* pB, int q, int swl, const LD& ld,
                                             char* nstage) {
  using Frag = typename Mma<T>::frag;
  static_assert(!LOADS || (LD::NP - 1) * ILV < 2 * TN, "every loader piece needs a tile to ride on");
  const int coff0 = ((0 * 4 + q) ^ swl) << 4, coff1 = ((1 * 4 + q) ^ swl) << 4;
  Frag fa[2][TM];
  Frag fb[2 * TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const Frag*>(pA + i * 16 * IG_BKB + coff0);
  fb[0] = *reinterpret_cast<const Frag*>(pB + coff0);
  fb[1] = *reinterpret_cast<const Frag*>(pB + 16 * IG_BKB + coff0);
  __builtin_amdgcn_sched_group_barrier(0x100, TM + 2, 0);
  dma_mma_tile<T, TM, TN, 0, LOADS, ILV, LD>(acc, fa, fb, pA, pB, coff0, coff1, ld, nstage);
}

// NW = 8: waves 4 (M) x 2 (N).  NW = 4: waves 2 x 2, ONE wave per SIMD with a (BM/2) x (BN/2) tile -- the fragment
// reads of a K step drop from 8 x (TM + TN) KB to 4 x (2 TM + TN) KB.  The stamps showed the 8-wave 192 x 208 step
// bound by LDS bandwidth (160 KB of fragment reads + 53 KB of DMA writes per step = 1664 clocks at 128 B/clk against
// 1344 clocks of MFMA), the second wave of each SIMD finishing its MFMAs 48 % later than the first.
// WS = 1 (wave-specialised, NW = 8): waves 0..3 only read fragments and issue MFMAs, each on (BM/4) rows x ALL BN
// columns (13 column tiles: no spill tile, 3 + 13 fragment KB per 39 MFMAs), waves 4..7 (one per SIMD beside a
// compute wave) only issue the LDS-DMA pieces.  A lone compute wave per SIMD (NW = 4) lost its MFMA issue slots
// whenever it sat in the issue of a 1 KB DMA piece; a 2 x 2 compute layout (6 x 7 tiles) does not fit 256 VGPRs.
template <typename T, int BM, int BN, int NS, int NW = 8, int WS = 0>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) igemm_fwd_dma_kernel(const IgArgs args) {
  static_assert(NS == 2 || NS == 3, "LDS ring depth");
  static_assert(NW == 8 || NW == 4, "wave layouts");
  static_assert(!WS || (NW == 8 && NS == 3), "the specialised form is the 8-wave 3-stage kernel");
  constexpr int CW = WS ? NW / 2 : NW;              // waves that compute
  constexpr int LW = WS ? NW / 2 : NW;              // waves that load
  static_assert(BM % (8 * LW) == 0 && BM % (CW * 8) == 0, "wave rows of k*16 rows, 8-row DMA instructions dealt in pairs");
  using LD = DmaLoader<T, BM, BN, LW>;
  constexpr int NTHR = 64 * NW;
  constexpr int WNW = WS ? 1 : 2;                   // wave columns
  constexpr int WMW = CW / WNW;                     // wave rows
  constexpr int NAJ = LD::NAJ;                      // A wave-instructions per wave per step
  constexpr int WMR = BM / WMW;                     // rows per wave row
  // waves 4 (M) x 2 (N): 64 rows x 7 or 6 column tiles.  Waves w and w+4 land on the same SIMD
  // (cyclic placement), so each SIMD carries 13 column tiles in total.
  constexpr int TM = WMR / 16, TN = WS ? BN / 16 : (BN / 16 + 1) / 2, TN1 = WS ? TN : BN / 16 - TN;
  constexpr int BNL = LD::BNL;                      // B rows kept in LDS: both wave columns run TN tiles (the
                                                    // second one's last tile may spill past BN and is discarded)
  constexpr int NIB = LD::NIB;                      // B wave-instructions per step (8 rows each)
  constexpr int NBJ = LD::NBJ;                      // per wave
  constexpr int STAGE = (BM + BNL) * IG_BKB;
  // tiles between loader pieces: a 2-stage ring wants them early, a 3-stage ring spread over the step
  constexpr int ILV = NS == 2 ? 1 : ((2 * ((BN / 16 + 1) / 2) - 1) / (LD::NP - 1) >= 2 ? 2 : 1);
  static_assert(BN % 16 == 0, "tile config");
  using Frag = typename Mma<T>::frag;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* sTap = reinterpret_cast<int*>(smem + NS * STAGE);

  const IgClass& cl = args.cls[blockIdx.z];
  const int tid = threadIdx.x;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = bid / args.ntiles_n;
  const int tile_n = bid - tile_m * args.ntiles_n;
  const int M = cl.M;
  if (tile_m * BM >= M) {                              // a class with fewer rows than the largest: its partial rows are zeros
    if (args.col_partial && tid < BN) {
      const int n = args.n_begin + tile_n * BN + tid;
      float* pr = args.col_partial + (size_t)(blockIdx.z * args.ntiles_m_max + tile_m) * 2 * args.N;
      if (n < args.N) { pr[n] = 0.f; pr[args.N + n] = 0.f; }
    }
    return;
  }
  const int m0 = tile_m * BM, n0 = args.n_begin + tile_n * BN;
  if (args.ksplit > 1 && (int)blockIdx.y * args.steps_per_split >= cl.nsteps) {
    // a class with fewer K steps than the largest: this split's partial tile is zeros
    float* sl = args.slab + ((size_t)(blockIdx.y * args.nclasses + blockIdx.z) * args.slab_rows + m0) * args.N;
    for (int i = tid; i < BM * BN; i += NTHR) {
      const int r = i / BN, c = i - r * BN;
      if (m0 + r < M && n0 + c < args.N) sl[(size_t)r * args.N + n0 + c] = 0.f;
    }
    return;
  }

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const bool is_loader = !WS || wave >= CW, is_mma = !WS || wave < CW;   // wave-uniform
  const int lwave = WS ? (wave - CW) & (LW - 1) : wave;
  const int par = lwave & 1, wh = lwave >> 1;
  const int rsub = lane >> 3;
  const int dbg = args.debug;                        // diagnostics: 4 / 5 / 6 = zero-record descriptor for A / B / both
  unsigned long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0;   // phase stamps (diagnostic build only)
  TDG_STAMP(ph0);
  const float bias_v = (tid < LD::BNL && args.bias && n0 + tid < args.N) ? args.bias[n0 + tid] : 0.f;

  LD ld;
  ld.rA = make_rsrc(args.src, (dbg == 4 || dbg == 6) ? 0u : args.src_bytes);
  ld.rB = make_rsrc(static_cast<const char*>(args.wpack) + cl.w_off_bytes, (dbg == 5 || dbg == 6) ? 0u : args.w_bytes - cl.w_off_bytes);
  ld.SH = args.SH; ld.SW = args.SW; ld.Cs = args.Cs; ld.CV = (int)args.fd_c.d;
  ld.ntaps = cl.ntaps; ld.tapreg = cl.tap[lane & (IG_MAX_TAPS - 1)];
  ld.fd_ck = args.fd_ck; ld.fd_nt = cl.fd_nt; ld.nslices = args.nslices;
  ld.lch = (lane & 7) ^ ((4 * par + (rsub >> 1)) & 7);   // this lane's logical K chunk

  // ---- A rows served by this lane: LDS rows 8*I_j + rsub, I_j = 2*(NAJ*wh + j) + par ------------------
#pragma unroll
  for (int j = 0; j < NAJ; ++j) {
    const int I = 2 * (NAJ * wh + j) + par;
    const int m = m0 + 8 * I + rsub;
    const bool ok = m < M;
    const unsigned mm = ok ? (unsigned)m : 0u;
    const unsigned nb = fd_div(mm, cl.fd_ghw);
    const unsigned rem = mm - nb * (unsigned)(cl.GH * cl.GW);
    const unsigned a = fd_div(rem, cl.fd_gw);
    const unsigned b = rem - a * (unsigned)cl.GW;
    ld.a_h[j] = ok ? (int)a * args.sigma : -(1 << 20);
    ld.a_w[j] = (int)b * args.sigma;
    ld.a_base[j] = ((nb * (unsigned)args.SH + a * (unsigned)args.sigma) * (unsigned)args.SW + b * (unsigned)args.sigma) * (unsigned)args.Cs;
    ld.a_lds[j] = I * 1024;
  }
  // ---- B rows: instruction I = 2*(wh + 4j) + par covers filter rows 8*I + rsub ---------------------------
#pragma unroll
  for (int j = 0; j < NBJ; ++j) {
    const int I = 2 * (wh + (LW / 2) * j) + par;
    const int n = n0 + 8 * I + rsub;
    // pieces past the tile's BN rows (the spill rows of the second wave column, whose products are discarded, and
    // the dummies that even out the piece count) load nothing: out-of-range source, destination inside the spill rows
    static_assert(NIB % LW == 0 || BNL > BN, "dummy pieces need spill rows to land in");
    const bool useful = 8 * I < BN;                    // wave-uniform
    ld.b_row[j] = (useful && n < args.N) ? ((unsigned)n * (unsigned)cl.Kp + (unsigned)(ld.lch * LD::VEC)) * (unsigned)sizeof(T) : OOB_OFFSET;
    ld.b_lds[j] = BM * IG_BKB + (I < NIB ? I : NIB - 1) * 1024;
  }

  // the tile's bias row goes to LDS: requested at the top of the kernel, written here (the row tables above covered
  // its latency), published by the first barrier of the K loop -- instead of TN dependent loads at the head of the
  // epilogue (one workgroup per CU: nothing else runs there)
  float* sBias = reinterpret_cast<float*>(sTap + IG_MAX_TAPS);
  if (tid < BNL) sBias[tid] = bias_v;

  const int r16 = lane & 15, q = lane >> 4;
  const int swl = (r16 >> 1) & 7;
  const int wm = wave % WMW, wn = wave / WMW;
  const int tnw = wn == 0 ? TN : TN1;               // wave-uniform
  static_assert((TN * 16) % 16 == 0 && ((TN * 16) >> 1) % 8 == 0, "swizzle term must not depend on the wave's column base");

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int s0 = (int)blockIdx.y * args.steps_per_split;      // split-K: this workgroup's K steps [s0, s0 + nsteps)
  const int nsteps = min(cl.nsteps - s0, args.steps_per_split);
  ld.step0 = s0;
  if constexpr (NS == 2) {
    ld.prepare(0);
    ld.all_pieces(smem);
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, s_issue = 0, s_mma = 0, s_sync = 0;
    for (int step = 0; step < nsteps - 1; ++step) {
      const int cur = step & 1;
      TDG_STAMP(t0);
      ld.prepare(step + 1);
      TDG_STAMP(t1);
      const char* pA = smem + cur * STAGE + (wm * WMR + r16) * IG_BKB;
      const char* pB = smem + cur * STAGE + BM * IG_BKB + (wn * TN * 16 + r16) * IG_BKB;
      char* nstage = smem + (cur ^ 1) * STAGE;
      dma_mma_step<T, TM, TN, true, ILV, LD>(acc, pA, pB, q, swl, ld, nstage);
      TDG_STAMP(t2);
      __syncthreads();
      TDG_STAMP(t3);
      s_issue += t1 - t0; s_mma += t2 - t1; s_sync += t3 - t2;
    }
    {
      const int cur = (nsteps - 1) & 1;
      const char* pA = smem + cur * STAGE + (wm * WMR + r16) * IG_BKB;
      const char* pB = smem + cur * STAGE + BM * IG_BKB + (wn * TN * 16 + r16) * IG_BKB;
      dma_mma_step<T, TM, TN, false, ILV, LD>(acc, pA, pB, q, swl, ld, smem);
      __syncthreads();
    }
#ifdef TDG_STAMPS
    if (args.stamps && lane == 0) {
      unsigned long long* o = args.stamps + ((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * 8 + wave) * 4;
      o[0] = s_issue; o[1] = s_mma; o[2] = s_sync; o[3] = (unsigned long long)(nsteps - 1);
    }
#endif
  } else {
    // 3-stage ring: the loads of step s+2 ride on the MFMAs of step s and stay in flight across the
    // barrier that ends it; a wave waits (counted vmcnt) only for its own pieces of step s+1 before that
    // barrier.  Raw s_barrier: __syncthreads() would drain every outstanding LDS-DMA.
    auto wait_barrier = [&](bool more_in_flight) {
      if (more_in_flight) {
        static_assert(LD::NP >= 3 && LD::NP <= 14, "add the immediate");
        if constexpr (LD::NP == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        if constexpr (LD::NP == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        if constexpr (LD::NP == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
        if constexpr (LD::NP == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        if constexpr (LD::NP == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        if constexpr (LD::NP == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        if constexpr (LD::NP == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        if constexpr (LD::NP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if constexpr (LD::NP == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        if constexpr (LD::NP == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        if constexpr (LD::NP == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        if constexpr (LD::NP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    };
    {
      // both steps' tap lookups before the first DMA: a DS instruction behind a pending LDS-DMA gets a vmcnt(0)
      LD l1 = ld;
      if (is_loader) {
        ld.prepare(0);
        l1.prepare(1);                               // (past the last step: every piece out of range)
        ld.all_pieces(smem);
        if (nsteps > 1) l1.all_pieces(smem + STAGE);
      }
    }
    wait_barrier(nsteps > 1);
    int cur = 0, step = 0;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, s_issue = 0, s_mma = 0, s_sync = 0;
    if constexpr (WS) {
      // two separate loops with the same barrier sequence: the loader state is dead in the compute loop (one merged
      // loop kept it live beside the accumulators and the fragments: spills)
      if (is_loader) {
        for (; dbg == 7 && step < nsteps - 2; ++step) {   // diagnostics: what the loop costs WITHOUT the gathered operand's pieces
          const int nxt2 = cur == 0 ? 2 : cur - 1;       // (results are garbage: the A rows of the ring are never refreshed)
          ld.prepare(step + 2);
          ld.b_pieces(smem + nxt2 * STAGE);
          static_assert(LD::NBJ == 7 || LD::NBJ == 4 || LD::NBJ == 8 || LD::NBJ == 2 || LD::NBJ == 1, "add the immediate");
          if constexpr (LD::NBJ == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          cur = cur == 2 ? 0 : cur + 1;
        }
        for (; step < nsteps - 2; ++step) {
          const int nxt2 = cur == 0 ? 2 : cur - 1;     // (cur + 2) % 3
          TDG_STAMP(t0);
          ld.prepare(step + 2);
          ld.all_pieces(smem + nxt2 * STAGE);
          TDG_STAMP(t2);
          wait_barrier(true);
          TDG_STAMP(t3);
          s_issue += t2 - t0; s_sync += t3 - t2;
          cur = cur == 2 ? 0 : cur + 1;
        }
        for (; step < nsteps; ++step) wait_barrier(false);
      } else {
        for (; step < nsteps; ++step) {
          const char* pA = smem + cur * STAGE + (wm * WMR + r16) * IG_BKB;
          const char* pB = smem + cur * STAGE + BM * IG_BKB + (wn * TN * 16 + r16) * IG_BKB;
          TDG_STAMP(t1);
          dma_mma_step<T, TM, TN, false, ILV, LD>(acc, pA, pB, q, swl, ld, smem);
          TDG_STAMP(t2);
          wait_barrier(step < nsteps - 2);
          TDG_STAMP(t3);
          s_mma += t2 - t1; s_sync += t3 - t2;
          cur = cur == 2 ? 0 : cur + 1;
        }
      }
#ifdef TDG_STAMPS
      if (args.stamps && lane == 0) {
        unsigned long long* o = args.stamps + ((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * 8 + wave) * 4;
        o[0] = s_issue; o[1] = s_mma; o[2] = s_sync; o[3] = (unsigned long long)(is_loader ? (nsteps > 2 ? nsteps - 2 : 1) : nsteps);
      }
#endif
    }
    for (; !WS && step < nsteps - 2; ++step) {
      const int nxt2 = cur == 0 ? 2 : cur - 1;         // (cur + 2) % 3
      TDG_STAMP(t0);
      ld.prepare(step + 2);                            // (worked out among the MFMAs of the previous step instead, this
      TDG_STAMP(t1);                                   //  block's 9 % of a step only moved into the MFMA phase: measured)
      const char* pA = smem + cur * STAGE + (wm * WMR + r16) * IG_BKB;
      const char* pB = smem + cur * STAGE + BM * IG_BKB + (wn * TN * 16 + r16) * IG_BKB;
      dma_mma_step<T, TM, TN, true, ILV, LD>(acc, pA, pB, q, swl, ld, smem + nxt2 * STAGE);
      TDG_STAMP(t2);
      wait_barrier(true);
      TDG_STAMP(t3);
      s_issue += t1 - t0; s_mma += t2 - t1; s_sync += t3 - t2;
      cur = cur == 2 ? 0 : cur + 1;
    }
#ifdef TDG_STAMPS
    if (!WS && args.stamps && lane == 0) {
      unsigned long long* o = args.stamps + ((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * 8 + wave) * 4;
      o[0] = s_issue; o[1] = s_mma; o[2] = s_sync; o[3] = (unsigned long long)(nsteps > 2 ? nsteps - 2 : 0);
    }
#endif
    for (; !WS && step < nsteps; ++step) {
      const char* pA = smem + cur * STAGE + (wm * WMR + r16) * IG_BKB;
      const char* pB = smem + cur * STAGE + BM * IG_BKB + (wn * TN * 16 + r16) * IG_BKB;
      dma_mma_step<T, TM, TN, false, ILV, LD>(acc, pA, pB, q, swl, ld, smem);
      wait_barrier(false);
      cur = cur == 2 ? 0 : cur + 1;
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------------
  TDG_STAMP(ph1);
  const int N = args.N, Cso = args.Cso;
  T* out = static_cast<T*>(args.out);
  const T* msk = static_cast<const T*>(args.mask_src);
  const float* bias = args.bias;
  const int act = args.act, mmode = args.mask_mode, accum = args.accumulate;
  const float leak = args.leak;

  if constexpr (!WS) {
    if (args.ksplit > 1) {                           // split-K: raw f32 partial tile; splitk_finish_kernel does the rest
      float* sl = args.slab + ((size_t)(blockIdx.y * args.nclasses + blockIdx.z) * args.slab_rows) * N;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 16 + q * 4;
        if (j >= tnw || n >= N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int m = m0 + wm * WMR + i * 16 + r16;
          if (m >= M) continue;
          if (n + 3 < N) {
            *reinterpret_cast<f32x4*>(sl + (size_t)m * N + n) = acc[i][j];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < N) sl[(size_t)m * N + n + e] = acc[i][j][e];
          }
        }
      }
      return;
    }
  }

  if constexpr (sizeof(T) == 2) {
    // bf16: stage the tile through the (now idle) LDS ring so that global stores (and mask loads) are whole
    // 16-byte chunks of contiguous pixel rows instead of 8-byte pieces scattered over 16 pixels per instruction
    if (!accum && (N & 3) == 0) {                            // (N % 8 == 4: the last chunk of a row is an 8-byte piece)
      constexpr int PE = BN * 2 + 16;                          // row pitch: +16 B keeps the 8-byte writes spread over banks
      constexpr int CPR = BN * 2 / 16;                         // 16-byte chunks per row
      static_assert(BM * PE + BM * 8 <= NS * STAGE, "epilogue staging must fit the ring");
      char* sE = smem;
      long long* sPix = reinterpret_cast<long long*>(smem + BM * PE);
      if (tid < BM) {
        const int m = m0 + tid;
        long long p = -1;
        if (m < M) {
          const unsigned nb = fd_div((unsigned)m, cl.fd_ghw);
          const unsigned rem = (unsigned)m - nb * (unsigned)(cl.GH * cl.GW);
          const unsigned a = fd_div(rem, cl.fd_gw);
          const unsigned b = rem - a * (unsigned)cl.GW;
          p = (long long)(((size_t)(nb * (unsigned)args.OH + a * (unsigned)args.os + (unsigned)cl.oh0) * (unsigned)args.OW +
                           b * (unsigned)args.os + (unsigned)cl.ow0) * (size_t)Cso);
        }
        sPix[tid] = p;
      }
      f32x4 bvs[TN];                                           // bias pieces of the wave's columns (zeros past N / no bias)
#pragma unroll
      for (int j = 0; j < TN; ++j) bvs[j] = *reinterpret_cast<const f32x4*>(sBias + (wn * TN + j) * 16 + q * 4);
      // one dispatch on the activation code, then branch-free loops over the accumulators
      if (is_mma) dispatch_act(act, [&](auto tag) {
        constexpr int ACT = decltype(tag)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = (wn * TN + j) * 16 + q * 4;
          if (j < tnw) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
              f32x4 v = acc[i][j] + bvs[j];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = apply_act_c<ACT>(v[e], act, leak);
              *reinterpret_cast<bf16x4*>(sE + (wm * WMR + i * 16 + r16) * PE + col * 2) =
                  bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            }
          }
        }
      });
      __syncthreads();
      TDG_STAMP(ph2);
      const float mlow = mask_low(mmode, leak);
      // chunk c = tid + NTHR * it of the tile.  The mask pieces of ALL the thread's chunks are requested before the
      // first one is used (a load -> multiply -> store chain per chunk paid the memory latency NIT times over)
      constexpr int NIT = (BM * CPR + NTHR - 1) / NTHR;
      long long pp[NIT];
      bf16x8 mv[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = tid + NTHR * it;
        const int row = c / CPR, cc = c - row * CPR;
        const int n = n0 + cc * 8;
        long long p = c < BM * CPR ? sPix[row] : -1;
        if (n >= N) p = -1;
        pp[it] = p;
        mv[it] = bf16x8{(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f};
        if (mmode != TDG_MASK_NONE && p >= 0) {
          if (n + 8 <= N) {
            mv[it] = *reinterpret_cast<const bf16x8*>(msk + p + n);
          } else {                                             // 4 columns left
            const bf16x4 h = *reinterpret_cast<const bf16x4*>(msk + p + n);
            mv[it][0] = h[0]; mv[it][1] = h[1]; mv[it][2] = h[2]; mv[it][3] = h[3];
          }
        }
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const long long p = pp[it];
        if (p < 0) continue;
        const int c = tid + NTHR * it;
        const int row = c / CPR, cc = c - row * CPR;
        const int n = n0 + cc * 8;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(sE + row * PE + cc * 16);
        if (mmode != TDG_MASK_NONE) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] * ((float)mv[it][e] > 0.f ? 1.f : mlow));
          if (args.col_partial) *reinterpret_cast<bf16x8*>(sE + row * PE + cc * 16) = v;     // the sums are of what is stored
        }
        if (n + 8 <= N) {
          *reinterpret_cast<bf16x8*>(out + p + n) = v;
        } else {
          *reinterpret_cast<bf16x4*>(out + p + n) = bf16x4{v[0], v[1], v[2], v[3]};
        }
      }
      if (args.col_partial) {
        // Column partials of the stored tile (TdgEpilogue.col_partial): NTHR / BN thread groups walk interleaved rows of
        // one column each, the groups are summed in a fixed order (deterministic).  TDG_COL_BN: deviations from the bias.
        constexpr int G = NTHR / BN > 0 ? NTHR / BN : 1;
        float* sRed = reinterpret_cast<float*>(smem + BM * PE + BM * 8);
        static_assert(BM * PE + BM * 8 + 2 * G * BN * 4 <= NS * STAGE, "column partials must fit the ring");
        __syncthreads();
        const int mlim = args.col_images > 0 ? min(M, args.col_images * cl.GH * cl.GW) : M;
        const int rows_valid = max(0, min(BM, mlim - m0));
        const int col = tid % BN, grp = tid / BN;
        const bool bnm = args.col_mode == TDG_COL_BN;
        if (grp < G) {
          const float piv = bnm ? sBias[col] : 0.f;
          float s0 = 0.f, s1 = 0.f;
          for (int r = grp; r < rows_valid; r += G) {
            const float v = (float)*reinterpret_cast<const bf16_t*>(sE + r * PE + col * 2) - piv;
            s0 += v;
            s1 += v * v;
          }
          sRed[(grp * 2 + 0) * BN + col] = s0;
          sRed[(grp * 2 + 1) * BN + col] = s1;
        }
        __syncthreads();
        if (tid < BN && n0 + tid < N) {
          float t0 = 0.f, t1 = 0.f;
#pragma unroll
          for (int g = 0; g < G; ++g) { t0 += sRed[(g * 2 + 0) * BN + tid]; t1 += sRed[(g * 2 + 1) * BN + tid]; }
          float* pr = args.col_partial + (size_t)(blockIdx.z * args.ntiles_m_max + tile_m) * 2 * N;
          pr[n0 + tid] = t0;
          pr[N + n0 + tid] = t1;
        }
      }
#ifdef TDG_STAMPS
      if (args.stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TDG_STAMP(ph3);
        if (lane == 0) {
          unsigned long long* o = args.stamps + 262144 + ((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * 8 + wave) * 4;
          o[0] = ph0; o[1] = ph1; o[2] = ph2; o[3] = ph3;
        }
      }
#endif
      return;
    }
  }

  // direct form (f32 tiles, accumulating epilogues, N not a multiple of 8): lane owns pixel r16 x 4 channels
  // (not compiled into the wave-specialised form -- its 3 x 13 unrolled copies pass the unroll budget, the
  //  accumulators would then live in scratch -- which the host only launches for staged epilogues)
  if constexpr (WS) return;
  if (!is_mma) return;
  size_t pix[TM];
  bool okm[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WMR + i * 16 + r16;
    okm[i] = m < M;
    const unsigned mm = okm[i] ? (unsigned)m : 0u;
    const unsigned nb = fd_div(mm, cl.fd_ghw);
    const unsigned rem = mm - nb * (unsigned)(cl.GH * cl.GW);
    const unsigned a = fd_div(rem, cl.fd_gw);
    const unsigned b = rem - a * (unsigned)cl.GW;
    pix[i] = ((size_t)(nb * (unsigned)args.OH + a * (unsigned)args.os + (unsigned)cl.oh0) * (unsigned)args.OW +
              b * (unsigned)args.os + (unsigned)cl.ow0) * (size_t)Cso;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 16 + q * 4;
    const bool okn = (j < tnw) && (n < N);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(sBias + (wn * TN + j) * 16 + q * 4);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (okn && okm[i]) {
        f32x4 v = acc[i][j] + bv;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act, leak);
        if (accum) {
          if constexpr (sizeof(T) == 4) {
            v += *reinterpret_cast<const f32x4*>(out + pix[i] + n);
          } else {
            const bf16x4 ov = *reinterpret_cast<const bf16x4*>(out + pix[i] + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)ov[e];
          }
        }
        if (mmode != TDG_MASK_NONE) {
          if constexpr (sizeof(T) == 4) {
            const f32x4 mv = *reinterpret_cast<const f32x4*>(msk + pix[i] + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= mask_factor(mv[e], mmode, leak);
          } else {
            const bf16x4 mv = *reinterpret_cast<const bf16x4*>(msk + pix[i] + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= mask_factor((float)mv[e], mmode, leak);
          }
        }
        if constexpr (sizeof(T) == 4) {
          *reinterpret_cast<f32x4*>(out + pix[i] + n) = v;
        } else {
          *reinterpret_cast<bf16x4*>(out + pix[i] + n) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        }
      }
    }
  }
}

// ============================================================================================
// Patch-resident forward GEMM (igemm_fwd_patch_kernel): the gathered operand from LDS-resident patches.
//
// What bounds igemm_fwd_dma_kernel is the CU's L2 -> LDS intake (~35 B/clk): per K step its 192 x 208 tile takes in
// 24 KB of im2col rows and 27 KB of filter rows for 1248 clocks of MFMA.  The im2col rows are mostly re-reads: a 5x5
// stride-2 conv reads every input pixel for 6.25 taps, a stride-1 parity class of the backward-data GEMM for up to 9.
// Here a workgroup's row tile covers WHOLE images, the K order of the packed filter is channel-slice-major
// (IgArgs.fd_ck: 5 chunks = 40 channels per slice), and the source pixels a slice's taps read -- one sub-lattice of the
// source per tap group (IgClass.grp: the four pixel parities of a stride-2 forward conv, the whole image of a stride-1
// gather) -- are staged ONCE per (slice, group) "phase" as a patch of [image][qh][qw] 80-byte pixels (an odd number of
// 16-byte chunks: the fragment reads spread over all banks).  The A fragment of (row, K chunk) is then a ds_read_b128
// at  patch(phase) + (pixel(row) + dq(tap)) * 80 + j * 16,  or of a zero pixel when the tap falls outside the image:
// a per-chunk table in LDS (built at kernel start) holds the patch buffer, j, dq and the tap's lattice offset.
// Intake per K step drops to the filter rows plus ~4 KB of patch.
//
// 8 waves: 4 compute (48 rows x all 208 columns each, as igemm_fwd_dma_kernel's wave-specialised form) + 4 loaders.
// Per K step a loader wave issues the 7 filter pieces of step s + 2 into the 3-stage ring and, in the two steps in which a
// patch is loaded, PT_APS = 2 patch pieces; it waits for them (vmcnt(0)) before the NEXT step's barrier -- the one barrier of
// a step sits in the middle of the compute waves' step -- so a piece issued behind barrier M(s) is read from M(s + 1) on.
// (No filler pieces to equalise the waves' piece counts: issuing a 1 KiB piece costs a loader ~130 clocks whether or not it
// carries data.  The loaders have that time to spare today -- they wait 350 - 530 clocks per step -- and dropping the two
// out-of-tile filter pieces a step still carries changed nothing: DESIGN.md section 4, experiment (iii).)  Patches live in
// PT_NPB = 3 buffers: phase p is loaded (4 pieces per wave, over 2 steps) as soon as the last step that reads phase p - 3
// is over; the host only selects this kernel when every phase is then complete two steps before its first read
// (plan_fwd_patch).
// ============================================================================================
#define PT_CK 5                        // chunks per patch pixel
#define PT_PIXB (PT_CK * 16)
#define PT_NPB 3                       // patch buffers
#define PT_PIECES 16                   // 1 KiB pieces per patch, 4 per loader wave
#define PT_PATCHB (PT_PIECES * 1024)   // <= 204 pixels
#define PT_MAXPIX (PT_PATCHB / PT_PIXB)
#define PT_APS 2                       // patch pieces per loader wave and K step
#define PT_ZEROB 256                   // the zero pixel (LDS byte 0)
// an A fragment by its LDS byte address (the dynamic LDS block of this kernel starts at LDS byte 0 -- it declares no static
// LDS; the kernel checks -- so a table word IS the address: no per-read add of the block's base)
__device__ __forceinline__ bf16x8 pt_lds_frag(int byte_addr) {
  return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>((__attribute__((address_space(3))) char*)nullptr + byte_addr);
}

// the bf16 epilogue of a BM x BN tile staged through LDS (shared with the forms of igemm_fwd_dma_kernel that inline it):
// compute waves park bias + activation of their accumulators as bf16 rows, then every thread moves whole 16-byte
// chunks of pixel rows (mask multiply, column partials: see igemm_fwd_dma_kernel)
template <int BM, int BN, int TM, int TN, int NTHR>
__device__ __forceinline__ void staged_epilogue_bf16(f32x4 (&acc)[TM][TN], char* smem, const float* sBias, bool is_mma, int row0, int col_tile0,
                                                     int tnw, int tid, int r16, int q, int m0, int n0, int M, int tile_m,
                                                     const IgArgs& args, const IgClass& cl) {
  using T = bf16_t;
  const int N = args.N, Cso = args.Cso;
  T* out = static_cast<T*>(args.out);
  const T* msk = static_cast<const T*>(args.mask_src);
  const int act = args.act, mmode = args.mask_mode;
  const float leak = args.leak;
  constexpr int PE = BN * 2 + 16;
  constexpr int CPR = BN * 2 / 16;
  char* sE = smem;
  long long* sPix = reinterpret_cast<long long*>(smem + BM * PE);
  if (tid < BM) {
    const int m = m0 + tid;
    long long p = -1;
    if (m < M) {
      unsigned nb = fd_div((unsigned)m, cl.fd_ghw);
      const unsigned rem = (unsigned)m - nb * (unsigned)(cl.GH * cl.GW);
      unsigned a = fd_div(rem, cl.fd_gw);
      unsigned b = rem - a * (unsigned)cl.GW;
      if (cl.nbh * cl.nbw > 1) {                       // block tiles (igemm_fwd_patch_kernel): nb counts blocks, (a, b) are block-local
        const unsigned nblk = (unsigned)(cl.nbh * cl.nbw), img = nb / nblk, br = nb - img * nblk, bi = br / (unsigned)cl.nbw;
        a += bi * (unsigned)cl.GH;
        b += (br - bi * (unsigned)cl.nbw) * (unsigned)cl.GW;
        nb = img;
      }
      p = (long long)(((size_t)(nb * (unsigned)args.OH + a * (unsigned)args.os + (unsigned)cl.oh0) * (unsigned)args.OW +
                       b * (unsigned)args.os + (unsigned)cl.ow0) * (size_t)Cso);
    }
    sPix[tid] = p;
  }
  if (is_mma) {
    f32x4 bvs[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bvs[j] = *reinterpret_cast<const f32x4*>(sBias + (col_tile0 + j) * 16 + q * 4);
    dispatch_act(act, [&](auto tag) {
      constexpr int ACT = decltype(tag)::value;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = (col_tile0 + j) * 16 + q * 4;
        if (j < tnw) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            f32x4 v = acc[i][j] + bvs[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act_c<ACT>(v[e], act, leak);
            *reinterpret_cast<bf16x4*>(sE + (row0 + i * 16 + r16) * PE + col * 2) =
                bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          }
        }
      }
    });
  }
  __syncthreads();
  const float mlow = mask_low(mmode, leak);
  constexpr int NIT = (BM * CPR + NTHR - 1) / NTHR;
  long long pp[NIT];
  bf16x8 mv[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = tid + NTHR * it;
    const int row = c / CPR, cc = c - row * CPR;
    const int n = n0 + cc * 8;
    long long p = c < BM * CPR ? sPix[row] : -1;
    if (n >= N) p = -1;
    pp[it] = p;
    mv[it] = bf16x8{(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f};
    if (mmode != TDG_MASK_NONE && p >= 0) {
      if (n + 8 <= N) {
        mv[it] = *reinterpret_cast<const bf16x8*>(msk + p + n);
      } else {
        const bf16x4 h = *reinterpret_cast<const bf16x4*>(msk + p + n);
        mv[it][0] = h[0]; mv[it][1] = h[1]; mv[it][2] = h[2]; mv[it][3] = h[3];
      }
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const long long p = pp[it];
    if (p < 0) continue;
    const int c = tid + NTHR * it;
    const int row = c / CPR, cc = c - row * CPR;
    const int n = n0 + cc * 8;
    bf16x8 v = *reinterpret_cast<const bf16x8*>(sE + row * PE + cc * 16);
    if (mmode != TDG_MASK_NONE) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] * ((float)mv[it][e] > 0.f ? 1.f : mlow));
      if (args.col_partial) *reinterpret_cast<bf16x8*>(sE + row * PE + cc * 16) = v;
    }
    if (n + 8 <= N) {
      *reinterpret_cast<bf16x8*>(out + p + n) = v;
    } else {
      *reinterpret_cast<bf16x4*>(out + p + n) = bf16x4{v[0], v[1], v[2], v[3]};
    }
  }
  if (args.col_partial) {
    constexpr int G = NTHR / BN > 0 ? NTHR / BN : 1;
    float* sRed = reinterpret_cast<float*>(smem + BM * PE + BM * 8);
    __syncthreads();
    const int mlim = args.col_images > 0 ? min(M, args.col_images * cl.GH * cl.GW * (cl.nbh * cl.nbw > 1 ? cl.nbh * cl.nbw : 1)) : M;
    const int rows_valid = max(0, min(BM, mlim - m0));
    const int col = tid % BN, grp = tid / BN;
    const bool bnm = args.col_mode == TDG_COL_BN;
    if (grp < G) {
      const float piv = bnm ? sBias[col] : 0.f;
      float s0 = 0.f, s1 = 0.f;
      for (int r = grp; r < rows_valid; r += G) {
        const float v = (float)*reinterpret_cast<const bf16_t*>(sE + r * PE + col * 2) - piv;
        s0 += v;
        s1 += v * v;
      }
      sRed[(grp * 2 + 0) * BN + col] = s0;
      sRed[(grp * 2 + 1) * BN + col] = s1;
    }
    __syncthreads();
    if (tid < BN && n0 + tid < N) {
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) { t0 += sRed[(g * 2 + 0) * BN + tid]; t1 += sRed[(g * 2 + 1) * BN + tid]; }
      float* pr = args.col_partial + (size_t)(blockIdx.z * args.ntiles_m_max + tile_m) * 2 * N;
      pr[n0 + tid] = t0;
      pr[N + n0 + tid] = t1;
    }
  }
}

// One K step of a compute wave -- TM row tiles x TN column tiles x two 32-deep halves = 2 TN "tiles" of TM MFMAs -- as a
// chain that never drains: B fragment t + PT_LA is requested before the MFMAs of tile t issue, and the chain runs ACROSS the
// step boundary.  First half (tiles 0 .. TN-1, K chunks 0-3): A fragments fa0 (loaded during the previous step), second
// half's A fragments fa1 requested behind the first tiles.  Second half (after the step's "data ready" barrier): the NEXT
// step's fa0 are requested into the registers the first half has finished with, and the last PT_LA tiles request the next
// step's first B fragments from the next ring stage.  fb is indexed by tile; fb[2 TN ..] carry over.
// The sched_group_barrier pipeline (DS reads of a tile, then its MFMAs) hands out DS instructions in program order, so
// EVERY DS read of the loop body must have a slot: the chunk-table read in front of the first half has a group of its own
// (without it the table read took fb's slot and every later fragment moved one group back: 48 clocks from request to use
// instead of 96).  Measured with the slots right and PT_LA = 2, 3, 4 on one box: no change in time -- the step is not
// bound by fragment latency; see DESIGN.md section 4, "What the patch kernel is bound by" (power, not cycles).
#ifndef PT_VBLOCK
#define PT_VBLOCK 32
#endif
#ifndef PT_LA
#define PT_LA 3                        // B fragments requested this many tiles ahead of their MFMAs
#endif
template <int TM, int TN, int ABL, int t>
__device__ __forceinline__ void patch_half0(f32x4 (&acc)[TM][TN], const bf16x8 (&fa0)[TM], bf16x8 (&fa1)[TM], bf16x8 (&fb)[2 * TN + PT_LA],
                                            const char* smem, const int (&a1)[TM], const char* pB, int coff0, int coff1) {
  if constexpr (t < TN) {
    constexpr int t2 = t + PT_LA, ks2 = t2 / TN, j2 = t2 - ks2 * TN;
    if constexpr (ABL != 1) fb[t2] = *reinterpret_cast<const bf16x8*>(pB + j2 * 16 * IG_BKB + (ks2 ? coff1 : coff0));
    else fb[t2] = fb[t2 & 1];
    if constexpr (t < TM && ABL != 2) fa1[t] = pt_lds_frag(a1[t]);
#pragma unroll
    for (int i = 0; i < TM; ++i) Mma<bf16_t>::run(acc[i][t], fb[t], fa0[i]);
    __builtin_amdgcn_sched_group_barrier(0x100, 1 + (t < TM ? 1 : 0), 0);
    __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);
    patch_half0<TM, TN, ABL, t + 1>(acc, fa0, fa1, fb, smem, a1, pB, coff0, coff1);
  }
}
template <int TM, int TN, int ABL, int t>
__device__ __forceinline__ void patch_half1(f32x4 (&acc)[TM][TN], bf16x8 (&fa0)[TM], const bf16x8 (&fa1)[TM], bf16x8 (&fb)[2 * TN + PT_LA],
                                            const char* smem, const int (&a0n)[TM], const char* pB, const char* pBn, int coff0, int coff1) {
  constexpr int NT = 2 * TN;
  if constexpr (t < NT) {
    constexpr int j = t - TN, t2 = t + PT_LA;
    if constexpr (ABL == 1) fb[t2] = fb[t2 & 1];
    else if constexpr (t2 < NT) fb[t2] = *reinterpret_cast<const bf16x8*>(pB + (t2 - TN) * 16 * IG_BKB + coff1);
    else fb[t2] = *reinterpret_cast<const bf16x8*>(pBn + (t2 - NT) * 16 * IG_BKB + coff0);       // the next step's first tiles
    // the next step's A fragments into fa0 (tile TN no longer reads it): one per tile from the second tile on where the half has
    // more tiles than the wave has row tiles (208 / 128 columns), else spread over all of the half's tiles (64 columns: TN = TM)
    constexpr int a_lo = TN > TM ? (j >= 1 && j <= TM ? j - 1 : 0) : (j * TM) / TN;
    constexpr int a_hi = TN > TM ? (j >= 1 && j <= TM ? j : 0) : ((j + 1) * TM) / TN;
    if constexpr (ABL != 2) static_for<a_lo, a_hi>([&](auto i_c) { constexpr int i = decltype(i_c)::value; fa0[i] = pt_lds_frag(a0n[i]); });
#pragma unroll
    for (int i = 0; i < TM; ++i) Mma<bf16_t>::run(acc[i][j], fb[t], fa1[i]);
    __builtin_amdgcn_sched_group_barrier(0x100, 1 + (a_hi - a_lo), 0);
    __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);
    patch_half1<TM, TN, ABL, t + 1>(acc, fa0, fa1, fb, smem, a0n, pB, pBn, coff0, coff1);
  }
}

// ABL (diagnostic instantiations, TDG_PATCH_ABL): 1 = no B fragment reads in the loop, 2 = no A fragment reads, 3 = no address arithmetic,
// 4 = no LDS-DMA pieces in the loop
// CK: 16-byte chunks (8 channels) per K slice: 5 (the GAN's 200 / 400 / 800 channels) or 4 (32-channel slices: pix2pix / VAE
// widths; the fifth chunk of the 80-byte patch pixel is padding).  PIECES: 1 KiB pieces per patch buffer.  Row tiles are whole
// images or -- IgClass.nbh * nbw > 1 -- BLOCKS of one image (bh x bw anchors, rows in block-major order) whose patch carries a
// halo of `halo` lattice pixels on every side (out-of-image halo pixels are out-of-range sources = zeros: every tap of every
// row reads inside its patch).
template <int BM, int BN, int ABL, int CK = PT_CK, int PIECES = PT_PIECES>
__global__ void __launch_bounds__(512, 2) igemm_fwd_patch_kernel(const IgArgs args) {
  using T = bf16_t;
  constexpr int NTHR = 512, CW = 4, LW = 4;
  constexpr int PATCHB = PIECES * 1024, PPW = (PIECES + LW - 1) / LW, PPW0 = (PPW + 1) / 2;     // pieces per loader wave and patch; of them in the first of its two steps
  constexpr int WMR = BM / CW, TM = WMR / 16, TN = BN / 16;
  constexpr int BNL = 2 * ((BN / 16 + 1) / 2) * 16;      // filter rows kept per stage (whole 8-row piece pairs)
  constexpr int NIB = BNL / 8, NBJ = (NIB + LW - 1) / LW;
  constexpr int STAGE = BNL * IG_BKB, NS = 3;
  static_assert(PIECES % LW == 0, "patch pieces are dealt evenly to the loader waves");
  static_assert(BM % (16 * CW) == 0 && BN % 16 == 0 && BN / 16 >= BM / (16 * CW), "tile config: the K step hands out one A fragment per column tile");
  // LDS map: [zero pixel + dummy landing zones | filter ring | patches | chunk table | bias row].  The zero pixel sits at
  // byte 0, so "this tap is outside the image" is an AND of the fragment address with 0.
  constexpr int OFF_ZERO = 0, OFF_RING = PT_ZEROB, OFF_PATCH = OFF_RING + NS * STAGE, OFF_TAB = OFF_PATCH + PT_NPB * PATCHB;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const IgClass& cl = args.cls[blockIdx.z];
  const int tid = threadIdx.x;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = bid / args.ntiles_n;
  const int tile_n = bid - tile_m * args.ntiles_n;
  const int M = cl.M;
  const int nsteps = cl.nsteps;
  int* sTab = reinterpret_cast<int*>(smem + OFF_TAB);
  float* sBias = reinterpret_cast<float*>(smem + OFF_TAB + nsteps * 8 * 8);
  if (tile_m * BM >= M) {                              // a class with fewer rows than the largest: its partial rows are zeros
    if (args.col_partial && tid < BN) {
      const int n = args.n_begin + tile_n * BN + tid;
      float* pr = args.col_partial + (size_t)(blockIdx.z * args.ntiles_m_max + tile_m) * 2 * args.N;
      if (n < args.N) { pr[n] = 0.f; pr[args.N + n] = 0.f; }
    }
    return;
  }
  const int m0 = tile_m * BM, n0 = args.n_begin + tile_n * BN;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const bool is_mma = wave < CW;
  const int ntaps = cl.ntaps, G = cl.ngroups, QH = cl.QH, QW = cl.QW;
  const int GHW = cl.GH * cl.GW;
  const int SLC = ntaps * CK;                          // chunks per slice
  const int halo = cl.halo;
  const int P = args.nslices * G;                      // phases

  unsigned long long ph0 = 0, ph1 = 0, php = 0, t0 = 0, t1 = 0, t2 = 0, s_issue = 0, s_mma = 0, s_sync = 0;   // stamps (diagnostic build only)
  unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0;
  TDG_STAMP(ph0);
  // ---- tables: bias row, zero pixel, per-chunk A table ------------------------------------------------------------
  const float bias_v = (tid < BNL && args.bias && n0 + tid < args.N) ? args.bias[n0 + tid] : 0.f;
  if (tid < PT_ZEROB / 4) reinterpret_cast<int*>(smem + OFF_ZERO)[tid] = 0;
  if ((unsigned)(size_t)(lds_ptr_t)smem != 0u) __builtin_trap();      // (pt_lds_frag: table words are absolute LDS addresses)
  // lane t (< 32) holds tap t of the class: group, lattice offsets (+8, unsigned nibbles) -- read with lane crossbars below.
  // Built from SCALAR loads of the class's tap table (they ride in the kernel-argument batch at kernel entry; a per-lane
  // load of cl.tap[lane] was a cold vector-memory miss in front of everything the prologue does).  Compute waves only.
  TDG_STAMP(q0);
  int tapvec = 0;
  if (is_mma) {
    // every kernel-argument word first (two batches of scalar loads), then a fully unrolled loop over registers: a loop
    // with a dependent scalar load per tap cost ~300 clocks per iteration
    int tw[IG_MAX_TAPS / 2], g_t0[4], g_ph[4], g_pw[4];
    const int sgs = args.sigma - 1;                      // log2(sigma)
#pragma unroll
    for (int k = 0; k < IG_MAX_TAPS / 2; ++k) tw[k] = reinterpret_cast<const int*>(cl.tap)[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) { g_t0[k] = cl.grp[k].t0; g_ph[k] = cl.grp[k].ph; g_pw[k] = cl.grp[k].pw; }
    {
      // lane t picks its own tap word out of the 16 scalar registers (a select chain: ~30 instructions; one pass over the
      // taps with the lane as the loop variable was ~800)
      const int t = lane & (IG_MAX_TAPS - 1);
      int w = tw[0];
#pragma unroll
      for (int k = 1; k < IG_MAX_TAPS / 2; ++k) w = (t >> 1) == k ? tw[k] : w;
      int gi = 0, ph = g_ph[0], pw = g_pw[0];
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        const bool in = k < G && t >= g_t0[k];
        gi = in ? k : gi; ph = in ? g_ph[k] : ph; pw = in ? g_pw[k] : pw;
      }
      const int pk = (t & 1) ? (w >> 16) : w;
      const int dhq = (tap_dh(pk) - ph) >> sgs, dwq = (tap_dw(pk) - pw) >> sgs;          // exact multiples of sigma (1 or 2)
      tapvec = gi | ((dhq + 8) << 4) | ((dwq + 8) << 8);
    }
    TDG_STAMP(q1);
    // chunk table, two words per K chunk: the fragment's byte address minus the row's pixel offset (patch buffer, dq pixels,
    // chunk j) and the tap's index = the bit of the row's validity mask to test (31: K padding, never set)
    for (int g0 = 0; g0 < nsteps * 8; g0 += 64 * CW) {
      const int g = g0 + tid;
      const unsigned u = fd_div((unsigned)g, args.fd_ck);
      const int j = g - (int)u * CK;
      const unsigned sl = fd_div(u, cl.fd_nt);
      const int t = (int)(u - sl * (unsigned)ntaps);
      const int tv = __builtin_amdgcn_ds_bpermute((t & 31) << 2, tapvec);
      int w0 = 0, w1 = 31;
      if ((int)sl < args.nslices) {
        const int gi = tv & 15, dhq = ((tv >> 4) & 15) - 8, dwq = ((tv >> 8) & 15) - 8;
        const int p = (int)sl * G + gi;
        const int buf = p - (p / PT_NPB) * PT_NPB;
        w0 = OFF_PATCH + buf * PATCHB + (dhq * QW + dwq) * PT_PIXB + j * 16;
        w1 = t;
      }
      if (g < nsteps * 8) {
        sTab[2 * g] = w0;
        sTab[2 * g + 1] = w1;
      }
    }
  }
  TDG_STAMP(q2);
  if (tid < BNL) sBias[tid] = bias_v;
  TDG_STAMP(q3);

  const int r16 = lane & 15, q = lane >> 4;
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (!is_mma) {
    // ================================ loader waves ================================
    const int lwave = wave - CW;
    const int par = lwave & 1, wh = lwave >> 1, rsub = lane >> 3;
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(args.src, args.src_bytes);
    const __amdgpu_buffer_rsrc_t rB = make_rsrc(static_cast<const char*>(args.wpack) + cl.w_off_bytes, args.w_bytes - cl.w_off_bytes);
    const int lch = (lane & 7) ^ ((4 * par + (rsub >> 1)) & 7);       // this lane's logical K chunk of a filter row (source-side swizzle)
    unsigned b_row[NBJ];
    int b_lds[NBJ];
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
      const int I = 2 * (wh + (LW / 2) * j) + par;
      const int n = n0 + 8 * I + rsub;
      const bool useful = 8 * I < BN;
      b_row[j] = (useful && n < args.N) ? ((unsigned)n * (unsigned)cl.Kp + (unsigned)(lch * 8)) * 2u : OOB_OFFSET;
      b_lds[j] = (I < NIB ? I : NIB - 1) * 1024;
    }
    auto b_pieces = [&](int step) {                  // the filter rows of K step `step` into its ring stage (zero fill past the last step)
      const int st = step - (step / NS) * NS;
      char* stage = smem + OFF_RING + st * STAGE;
      const unsigned kb = step < nsteps ? (unsigned)step * IG_BKB : OOB_OFFSET;
#pragma unroll
      for (int j = 0; j < NBJ; ++j) {
        const unsigned off = (b_row[j] == OOB_OFFSET || kb == OOB_OFFSET) ? OOB_OFFSET : b_row[j] + kb;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr_t)(stage + b_lds[j]), 16, off, 0, 0, 0);
      }
    };
    // phase p = slice * G + group, walked incrementally: (slice, group, patch buffer) of the next phase to load and of
    // the phase PT_NPB behind it, whose last reading step frees that buffer
    struct Ph { int sl, gi, buf; };
    auto ph_next = [&](Ph& x) {
      if (++x.gi == G) { x.gi = 0; ++x.sl; }
      x.buf = x.buf == PT_NPB - 1 ? 0 : x.buf + 1;
    };
    auto ph_end_chunk = [&](const Ph& x) { return x.sl * SLC + (cl.grp[x.gi].t0 + cl.grp[x.gi].nt) * CK; };
    auto ph_delta = [&](const Ph& x) { return (unsigned)(((cl.grp[x.gi].ph * args.SW + cl.grp[x.gi].pw) * args.Cs + x.sl * CK * 8) * 2); };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's table / zero-pixel stores are done before any LDS-DMA is pending
    // ---- prologue: ring stage 0 first (its lane constants are cheap), the first patch, then what step 0 does not read yet
    // (patches 1 and 2, ring stage 1), which stays in flight across the prologue barrier: the wait of step 0 covers it
    b_pieces(0);
    // patch pieces: piece ps * 4 + lwave of a patch, 64 consecutive 16-byte chunks of its [pixel][PT_CK] image
    const int IPT = BM / GHW;                          // images (or blocks) per row tile
    const int nimg = M / GHW;
    const int nblk = cl.nbh * cl.nbw;                   // blocks per image (1: whole-image tiles)
    const int LH = args.SH / args.sigma, LW_ = args.SW / args.sigma;      // the source's sub-lattice
    unsigned a_src[PPW];
#pragma unroll
    for (int ps = 0; ps < PPW; ++ps) {
      const int c = (ps * LW + lwave) * 64 + lane;
      const int pl = c / PT_CK, j = c - pl * PT_CK;
      const unsigned il = fd_div((unsigned)pl, cl.fd_qhw);
      const int rem = pl - (int)il * QH * QW;
      const int qh = (int)fd_div((unsigned)rem, cl.fd_qw), qw = rem - qh * QW;
      const int blk = tile_m * IPT + (int)il;
      bool ok = (int)il < IPT && blk < nimg && j < CK;
      int img = blk, lh = qh, lw = qw;
      if (nblk > 1) {                                  // block tiles: lattice pixel = block origin - halo + (qh, qw)
        img = blk / nblk;
        const int br = blk - img * nblk, bi = br / cl.nbw, bj = br - bi * cl.nbw;
        lh = bi * cl.GH - halo + qh;
        lw = bj * cl.GW - halo + qw;
        ok = ok && (unsigned)lh < (unsigned)LH && (unsigned)lw < (unsigned)LW_;
      }
      a_src[ps] = ok ? (unsigned)((((img * args.SH + args.sigma * lh) * args.SW + args.sigma * lw) * args.Cs + j * 8) * 2) : OOB_OFFSET;
    }
    auto a_piece = [&](const Ph& x, unsigned delta, auto ps_c) {
      constexpr int ps = decltype(ps_c)::value;
      const unsigned off = a_src[ps] == OOB_OFFSET ? OOB_OFFSET : a_src[ps] + delta;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(smem + OFF_PATCH + x.buf * PATCHB + (ps * LW + lwave) * 1024), 16, off, 0, 0, 0);
    };
    auto a_pieces = [&](const Ph& x, unsigned delta, auto lo_c, auto hi_c) {       // pieces [lo, hi) of this wave's share of a patch
      static_for<decltype(lo_c)::value, decltype(hi_c)::value>([&](auto ps_c) { a_piece(x, delta, ps_c); });
    };
    Ph pn{0, 0, 0};
    {
      const unsigned dl0 = ph_delta(pn);
      a_pieces(pn, dl0, IntC<0>{}, IntC<PPW>{});
      ph_next(pn);
    }
    b_pieces(1);
    TDG_STAMP(t0);
    s_mma = t0 - ph0;                                    // (stamps: kernel entry -> prologue pieces issued)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBJ) : "memory");       // stage 0 and patch 0 have landed; stage 1 stays in flight
    __builtin_amdgcn_s_barrier();
    TDG_STAMP(php);
    // Patches 1 and 2 are loaded whole (4 pieces per wave) in steps 0 and 1 -- their buffers are free, and with every
    // workgroup of a round in its prologue at once the first loads come from HBM at ~11 B/clk/CU: the prologue only waits for
    // what step 0 reads.  From patch PT_NPB on: two pieces per step once the buffer's previous phase has been read.
    Ph pf{0, 0, 0};                                      // the phase whose buffer pn takes over (from patch PT_NPB on)
    int left = P - 1;                                    // phases still to load
    int ready = P > PT_NPB ? (ph_end_chunk(pf) - 1) / 8 + 1 : (1 << 30);
    if (ready < PT_NPB - 1) ready = PT_NPB - 1;          // (steps 0 .. PT_NPB - 2 carry the whole-patch loads)
    unsigned dl = left > 0 ? ph_delta(pn) : 0u;
    // ONE barrier per K step, in the middle of the compute waves' step (between its two 32-deep halves): behind barrier M(s)
    // the loaders overwrite ring stage s + 2 (= s - 1: read for the last time before M(s)) and wait for those pieces to land
    // before M(s + 1), behind which the compute waves start reading them.  A loader has a whole step for ~550 clocks of issue
    // and the load latency; the compute waves never wait for a second barrier.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // ring stage 1 (requested in the prologue)
    for (int step = 0; step < nsteps; ++step) {
      TDG_STAMP(t0);
      __builtin_amdgcn_s_barrier();                        // M(step)
      TDG_STAMP(t1);
      if constexpr (ABL != 4) b_pieces(step + 2);
      if constexpr (ABL == 4) {
      } else if (step < PT_NPB - 1) {
        if (left > 0) {
          a_pieces(pn, dl, IntC<0>{}, IntC<PPW>{});
          ph_next(pn);
          --left;
          dl = left > 0 ? ph_delta(pn) : 0u;
        }
      } else if (step == ready) {
        a_pieces(pn, dl, IntC<0>{}, IntC<PPW0>{});
      } else if (step == ready + 1) {
        a_pieces(pn, dl, IntC<PPW0>{}, IntC<PPW>{});
        ph_next(pn);
        ph_next(pf);
        --left;
        ready = left > 0 ? max((ph_end_chunk(pf) - 1) / 8 + 1, step + 1) : (1 << 30);
        dl = left > 0 ? ph_delta(pn) : 0u;
      }
      TDG_STAMP(t2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      s_sync += t1 - t0; s_issue += t2 - t1;
    }
    __builtin_amdgcn_s_barrier();                          // F: the compute waves are done with ring and patches
  } else {
    // ================================ compute waves ================================
    // per row tile: the row's pixel offset inside a patch and a validity bit per tap (bit t: tap t reads inside the image)
    int base_i[TM], mask_i[TM];
    {
      int a_i[TM], b_i[TM];
      unsigned rowbits[TM], colbits[TM];
      bool ok_i[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int loc = wave * WMR + i * 16 + r16;
        ok_i[i] = m0 + loc < M;
        const unsigned il = fd_div((unsigned)loc, cl.fd_ghw);
        const int rem = loc - (int)il * GHW;
        a_i[i] = (int)fd_div((unsigned)rem, cl.fd_gw);
        b_i[i] = rem - a_i[i] * cl.GW;
        base_i[i] = (((int)il * QH + a_i[i] + halo) * QW + b_i[i] + halo) * PT_PIXB;
        rowbits[i] = colbits[i] = 0u;
      }
      // validity mask = (taps whose row offset keeps the row inside the lattice) & (the same for columns): per distinct
      // offset d one ballot over the tap lanes gives the set of taps with that offset (a loop over the taps cost ~3000 clocks)
      const bool is_tap = lane < 32 && lane < ntaps;
      const int my_dh = ((tapvec >> 4) & 15) - 8, my_dw = ((tapvec >> 8) & 15) - 8;
      for (int d = -8; d < 8; ++d) {
        const unsigned tr = (unsigned)__builtin_amdgcn_ballot_w64(is_tap && my_dh == d);
        const unsigned tc = (unsigned)__builtin_amdgcn_ballot_w64(is_tap && my_dw == d);
        if (tr | tc) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            rowbits[i] |= (unsigned)(a_i[i] + halo + d) < (unsigned)QH ? tr : 0u;
            colbits[i] |= (unsigned)(b_i[i] + halo + d) < (unsigned)QW ? tc : 0u;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) mask_i[i] = ok_i[i] ? (int)(rowbits[i] & colbits[i]) : 0;
    }
    const int swl = (r16 >> 1) & 7;
    const int coff0 = ((0 * 4 + q) ^ swl) << 4, coff1 = ((1 * 4 + q) ^ swl) << 4;
    // fragment byte address: (row offset + chunk word 0) where the tap is inside the image, 0 (the zero pixel) elsewhere:
    // a signed 1-bit field extract of the validity mask is the AND mask -- three plain vector instructions, no branch
    auto a_offsets = [&](const i32x2& e, int (&o)[TM]) {
#pragma unroll
      for (int i = 0; i < TM; ++i) o[i] = (base_i[i] + e[0]) & __builtin_amdgcn_sbfe(mask_i[i], e[1], 1);
    };
    const i32x2* sTab2 = reinterpret_cast<const i32x2*>(sTab);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    TDG_STAMP(t0);
    s_issue = t0 - ph0;                                    // (stamps: kernel entry -> tables and masks built)
    __builtin_amdgcn_s_barrier();                          // tables, zero pixel, first patches, ring stage 0
    TDG_STAMP(php);
    int a0[TM], a1[TM];
    a_offsets(sTab2[q], a0);
    a_offsets(sTab2[4 + q], a1);
    i32x2 e0 = sTab2[(nsteps > 1 ? 8 : 0) + q], e1 = sTab2[(nsteps > 1 ? 8 : 0) + 4 + q];     // step 1's chunks
    bf16x8 fa0[TM], fa1[TM];
    bf16x8 fb[2 * TN + PT_LA];
    {
      const char* pB = smem + OFF_RING + r16 * IG_BKB;
#pragma unroll
      for (int i = 0; i < TM; ++i) fa0[i] = pt_lds_frag(a0[i]);
#pragma unroll
      for (int k = 0; k < PT_LA; ++k) fb[k] = *reinterpret_cast<const bf16x8*>(pB + k * 16 * IG_BKB + coff0);
    }
    int st = 0;
    for (int step = 0; step < nsteps; ++step) {
      TDG_STAMP(t0);
      const int stn = st == NS - 1 ? 0 : st + 1;
      const char* pB = smem + OFF_RING + st * STAGE + r16 * IG_BKB;
      const char* pBn = smem + OFF_RING + stn * STAGE + r16 * IG_BKB;
      // the next step's fragment addresses (from the chunk words read a step ago) and the chunk words of the step after
      int a0n[TM], a1n[TM];
      if constexpr (ABL == 3) {
#pragma unroll
        for (int i = 0; i < TM; ++i) { a0n[i] = a0[i]; a1n[i] = a1[i]; }
      } else {
        a_offsets(e0, a0n);
        a_offsets(e1, a1n);
      }
      const int s2 = step + 2 < nsteps ? step + 2 : step;
      e0 = sTab2[s2 * 8 + q];
      e1 = sTab2[s2 * 8 + 4 + q];
      // every vector instruction of the step in ONE block in front of the pipeline: vector instructions between the MFMA groups
      // cost far more than in a block (measured both ways: DESIGN.md section 4)
      __builtin_amdgcn_sched_group_barrier(0x002, PT_VBLOCK, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // the chunk-table read (one ds_read2_b64): see patch_half0
      patch_half0<TM, TN, ABL, 0>(acc, fa0, fa1, fb, smem, a1, pB, coff0, coff1);
      TDG_STAMP(t1);
      __builtin_amdgcn_s_barrier();                      // M(step): the pieces of ring stage step + 1 and of the patches behind it have landed
      TDG_STAMP(t2);
      patch_half1<TM, TN, ABL, TN>(acc, fa0, fa1, fb, smem, a0n, pB, pBn, coff0, coff1);
#pragma unroll
      for (int k = 0; k < PT_LA; ++k) fb[k] = fb[2 * TN + k];
#pragma unroll
      for (int i = 0; i < TM; ++i) a1[i] = a1n[i];
      st = stn;
      s_sync += t2 - t1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // F: every wave has read its last fragments (the epilogue stages over ring and patches)
  }
  TDG_STAMP(ph1);
#ifdef TDG_STAMPS
  if (args.stamps && lane == 0) {
    unsigned long long* o = args.stamps + ((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * 8 + wave) * 4;
    o[0] = s_issue; o[1] = s_mma; o[2] = s_sync; o[3] = (unsigned long long)nsteps;
  }
#endif

  // ---- epilogue (every DMA has landed: the loaders' last wait is vmcnt(0)) ----------------------------------------------
  static_assert(BM * (BN * 2 + 16) + BM * 8 + 2 * (NTHR / BN) * BN * 4 <= OFF_TAB, "epilogue staging must not reach the tables");   // (it overlays zero pixel, ring and patches)
  staged_epilogue_bf16<BM, BN, TM, TN, NTHR>(acc, smem, sBias, is_mma, wave * WMR, 0, TN, tid, r16, q, m0, n0, M, tile_m, args, cl);
#ifdef TDG_STAMPS
  if (args.stamps) {
    unsigned long long ph3 = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TDG_STAMP(ph3);
    if (lane == 0) {
      unsigned long long* o = args.stamps + 262144 + ((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * 8 + wave) * 4;
      o[0] = ph0; o[1] = php; o[2] = ph1; o[3] = ph3;       // entry, end of prologue, end of K loop, end of epilogue
      unsigned long long* o2 = args.stamps + 524288 + ((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * 8 + wave) * 4;
      o2[0] = q0 - ph0; o2[1] = q1 - ph0; o2[2] = q2 - ph0; o2[3] = q3 - ph0;
    }
  }
#endif
}

// ============================================================================================
// Block-patch forward GEMM with EVERY wave loading and multiplying (igemm_fwd_bp_kernel<BN>, round 4).
//
// For the 4 x 4 / stride-2 layers of pix2pix (16 - 64 K steps, 128 / 64 columns).  Stamps of igemm_fwd_dma_kernel<256,128,3> on
// those shapes (tools/stamp_conv.py): a K step takes ~1 980 clocks for 1 024 of MFMA (issue 200, multiply 1 030 - 1 440, barrier
// 90 - 590, 160 unstamped), prologue ~8 000, epilogue ~7 000; the 4 + 4 specialised patch form has its four loader waves issue 30
// pieces a step at ~130 clocks each -- no slack.  Here, as in igemm_wgrad_patch_kernel:
//   * tile 256 rows = one 16 x 16-anchor block of one image, BN = 128 | 64 columns, eight waves as 4 (M) x 2 (N), 64 x BN/2 each;
//   * K in 32-channel slices (CK = 4: one MFMA k-slice = the four chunks of one (slice, tap) UNIT, a 64-deep step = two units), the
//     taps in groups of exactly four (the parity groups of a 4x4 stride-2 filter, the 2x2 taps of its backward-data classes): a phase
//     (slice, group) is two steps and reads ONE patch: the block's lattice window with a halo of one pixel, 18 x 18 pixels of 80 bytes;
//   * fragment address = per-lane constant (row's pixel, k-group) + the unit's offset (patch buffer, tap displacement) from an LDS
//     table: one v_add per A fragment, no masks (a tap outside the image reads a halo pixel that was an out-of-range source);
//   * 3-stage filter ring + 3 patch buffers; per step a wave issues NSLOT pieces -- the filter rows of step s+3 and half of the
//     patch of phase (s+4)/2 -- behind the step's skewed barrier, and waits for its own pieces of step s+1 with a counted vmcnt.
// The host selects it only for classes that meet these assumptions (plan_fwd_bp); everything else keeps its kernel.
// ============================================================================================
__device__ __forceinline__ void bp_dma(const i32x4& rsrc, unsigned voff, unsigned soff, unsigned lds_byte) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_byte) : "memory");
}
#define BP_PIECES 28
// pieces of a group that ride on the tail tiles of the step in front of it (the rest: on the next step's slice-0 tiles)
#define BP_NSLOT(BN) (((BN) / 8 + BP_PIECES / 2 + 7) / 8)
#define BP_NTAIL(BN) ((BN) / 32 - ((BN) / 64 > 0 ? (BN) / 64 : 1))
#define BP_NPB(BN) (BP_NSLOT(BN) / 2 < BP_NTAIL(BN) ? BP_NSLOT(BN) / 2 : BP_NTAIL(BN))
template <int BN>
__global__ void __launch_bounds__(512, 2) igemm_fwd_bp_kernel(const IgArgs args) {
  using T = bf16_t;
  constexpr int BM = 256, NTHR = 512, TM = 4, TN = BN / 32;
  constexpr int STAGE = BN * IG_BKB, NS = 3, PATCHB = BP_PIECES * 1024;
  constexpr int NBP = BN / 8;                              // filter pieces per step
  constexpr int NPH = BP_PIECES / 2;                       // patch pieces per step (a phase = two steps)
  constexpr int NSLOT = (NBP + NPH + 7) / 8;               // pieces per wave and step
  constexpr int OFF_RING = PT_ZEROB, OFF_PATCH = OFF_RING + NS * STAGE, OFF_DUMMY = OFF_PATCH + PT_NPB * PATCHB, OFF_TAB = OFF_DUMMY + 1024;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const IgClass& cl = args.cls[blockIdx.z];
  const int tid = threadIdx.x;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = bid / args.ntiles_n;
  const int tile_n = bid - tile_m * args.ntiles_n;
  const int M = cl.M, nsteps = cl.nsteps, ntaps = cl.ntaps, G = cl.ngroups, QW = cl.QW, QH = cl.QH, halo = cl.halo;
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
  int* sU = reinterpret_cast<int*>(smem + OFF_TAB);       // per unit: patch buffer + tap displacement (LDS byte offset)
  float* sBias = reinterpret_cast<float*>(smem + OFF_TAB + 2 * nsteps * 4);
  if (tile_m * BM >= M) {
    if (args.col_partial && tid < BN) {
      const int n = args.n_begin + tile_n * BN + tid;
      float* pr = args.col_partial + (size_t)(blockIdx.z * args.ntiles_m_max + tile_m) * 2 * args.N;
      if (n < args.N) { pr[n] = 0.f; pr[args.N + n] = 0.f; }
    }
    return;
  }
  const int m0 = tile_m * BM, n0 = args.n_begin + tile_n * BN;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wm = wave & 3, wn = wave >> 2;
  const int r16 = lane & 15, q = lane >> 4;
  const int sgs = args.sigma - 1;

  unsigned long long bs0 = 0, bs1 = 0, bs2 = 0, bs3 = 0, bw_wait = 0, bw_bar = 0;
  TDG_STAMP(bs0);
  // ---- tables: zero pixel, unit offsets, bias row -----------------------------------------------------------------
  if (tid < PT_ZEROB / 4) reinterpret_cast<int*>(smem)[tid] = 0;
  for (int u = tid; u < 2 * nsteps; u += NTHR) {
    const unsigned sl = fd_div((unsigned)u, cl.fd_nt);
    const int t = u - (int)sl * ntaps;
    int gi = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) gi = (k < G && t >= cl.grp[k].t0) ? k : gi;
    const int pk = cl.tap[t];
    const int dhq = (tap_dh(pk) - cl.grp[gi].ph) >> sgs, dwq = (tap_dw(pk) - cl.grp[gi].pw) >> sgs;
    const int ph = (int)sl * G + gi;
    sU[u] = (int)lds0 + OFF_PATCH + (ph % PT_NPB) * PATCHB + (dhq * QW + dwq) * PT_PIXB;
  }
  if (tid < BN) sBias[tid] = (args.bias && n0 + tid < args.N) ? args.bias[n0 + tid] : 0.f;

  // ---- loop-invariant fragment addresses ----------------------------------------------------------------------------
  unsigned abase[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int loc = wm * 64 + i * 16 + r16;
    const int a_ = (int)fd_div((unsigned)loc, cl.fd_gw), b_ = loc - a_ * cl.GW;            // (one block per tile: loc < GH * GW = 256)
    abase[i] = (unsigned)(((a_ + halo) * QW + b_ + halo) * PT_PIXB + q * 16);
  }
  const int swl = (r16 >> 1) & 7;
  const unsigned bb0 = lds0 + OFF_RING + (unsigned)((wn * TN * 16 + r16) * IG_BKB + (((0 * 4 + q) ^ swl) << 4));
  const unsigned bb1 = lds0 + OFF_RING + (unsigned)((wn * TN * 16 + r16) * IG_BKB + (((1 * 4 + q) ^ swl) << 4));

  // ---- the loader: slot k of wave w is piece w + 8 k of [NBP filter pieces | NPH patch pieces | dummies] ----------------
  const i32x4 rB = make_rsrc_words(static_cast<const char*>(args.wpack) + cl.w_off_bytes, args.w_bytes - cl.w_off_bytes);
  const i32x4 rA = make_rsrc_words(args.src, args.src_bytes);
  const i32x4 rZ = i32x4{rB[0], rB[1], 0, rB[3]};           // zero records: every lane out of range (steps past the end)
  unsigned vo[NSLOT], vo2[NSLOT];                          // per-lane source offsets (patch slots: first / second half of a patch)
  {
    const int nblk = cl.nbh * cl.nbw > 1 ? cl.nbh * cl.nbw : 1;
    const int img = tile_m / nblk, br = tile_m - img * nblk, bi = br / (cl.nbw > 0 ? cl.nbw : 1), bj = br - bi * (cl.nbw > 0 ? cl.nbw : 1);
    const int LH = args.SH >> sgs, LWd = args.SW >> sgs;
    static_for<0, NSLOT>([&](auto k_c) {
      constexpr int k = decltype(k_c)::value;
      const int P = wave + 8 * k;
      if (P < NBP) {
        const int row = P * 8 + (lane >> 3), n = n0 + row;
        const int lch = (lane & 7) ^ ((row >> 1) & 7);
        vo[k] = vo2[k] = n < args.N ? ((unsigned)n * (unsigned)cl.Kp + (unsigned)(lch * 8)) * 2u : OOB_OFFSET;
      } else {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int PP = hh * NPH + (P - NBP);
          const int c = PP * 64 + lane;
          const int pl = c / PT_CK, j = c - pl * PT_CK;
          const int qh = (int)fd_div((unsigned)pl, cl.fd_qw), qw = pl - qh * QW;
          const int lh = bi * cl.GH - halo + qh, lw = bj * cl.GW - halo + qw;
          const bool ok = P - NBP < NPH && j < 4 && pl < QH * QW && (unsigned)lh < (unsigned)LH && (unsigned)lw < (unsigned)LWd;
          const unsigned o = ok ? (unsigned)((((img * args.SH + args.sigma * lh) * args.SW + args.sigma * lw) * args.Cs + j * 8) * 2) : OOB_OFFSET;
          if (hh == 0) vo[k] = o; else vo2[k] = o;
        }
      }
    });
  }
  // group g (g >= 0): the filter rows of step g and the (g + 1) % 2 half of the patch of phase (g + 1) / 2  (phase 0 whole with group 0)
  auto piece = [&](auto k_c, int g, int st) {
    constexpr int k = decltype(k_c)::value;
    const int P = wave + 8 * k;
    if constexpr (8 * k + 7 < NBP) {                       // filter rows of step g into ring stage st
      const bool live = g < nsteps;
      bp_dma(live ? rB : rZ, vo[k], (unsigned)(g * IG_BKB), lds0 + (unsigned)(OFF_RING + st * STAGE + P * 1024));
    } else {
      const int php = (g + 1) >> 1, hh = (g + 1) & 1;      // phase and half
      const int sl = php / G, gi = php - sl * G;
      const bool real = P - NBP < NPH;                    // (a slot past the patch's pieces writes its zeros to a dummy KiB)
      const bool live = real && php < args.nslices * G;  // (a phase past the last one: zeros into a buffer nobody reads any more)
      const unsigned delta = live ? (unsigned)(((cl.grp[gi < G ? gi : 0].ph * args.SW + cl.grp[gi < G ? gi : 0].pw) * args.Cs + sl * 32) * 2) : 0u;
      const int buf = php % PT_NPB;
      const unsigned dst = lds0 + (unsigned)(real ? OFF_PATCH + buf * PATCHB + (hh * NPH + (P - NBP)) * 1024 : OFF_DUMMY);
      bp_dma(live ? rA : rZ, hh ? vo2[k] : vo[k], delta, dst);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 FA[2][TM], FB[2][TN];
  auto mma_tile = [&](auto ks_c, auto j_c) {
    constexpr int ks = decltype(ks_c)::value, j = decltype(j_c)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[ks][j], FA[ks][i], acc[i][j], 0, 0, 0);
  };
  auto readA = [&](auto ks_c, auto i_c, int uo) {
    constexpr int ks = decltype(ks_c)::value, i = decltype(i_c)::value;
    FA[ks][i] = pt_lds_frag((int)(abase[i] + (unsigned)uo));
  };
  auto readB = [&](auto st_c, auto ks_c, auto j_c) {
    constexpr int st = decltype(st_c)::value, ks = decltype(ks_c)::value, j = decltype(j_c)::value;
    FB[ks][j] = pt_lds_frag((int)((ks ? bb1 : bb0) + (unsigned)(st * STAGE + j * 16 * IG_BKB)));
  };

  // ---- prologue: groups 0, 1, 2 (filter steps 0 - 2; patch phase 0 whole, phase 1 whole) -------------------------------------
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();                                          // tables and the zero pixel are in LDS before any LDS-DMA is pending
  static_for<0, NSLOT>([&](auto k_c) { piece(k_c, 0, 0); });                 // filter 0 + patch 0 second half ...
  {                                                        // ... and patch 0 FIRST half: group "-1" carries it
    static_for<0, NSLOT>([&](auto k_c) {
      constexpr int k = decltype(k_c)::value;
      if constexpr (8 * k + 7 >= NBP) piece(k_c, -1, 0);
    });
  }
  static_for<0, NSLOT>([&](auto k_c) { piece(k_c, 1, 1); });
  static_for<0, BP_NPB(BN)>([&](auto k_c) { piece(k_c, 2, 2); });              // (the rest of group 2 rides on step 0's slice-0 tiles)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSLOT + BP_NPB(BN)) : "memory");   // group 0 and patch 0 have landed; groups 1, 2 stay in flight
  __syncthreads();
  TDG_STAMP(bs1);
  int uo0 = sU[0], uo1 = sU[1];                             // unit offsets of the step being multiplied
  static_for<0, TM>([&](auto i_c) { readA(IntC<0>{}, i_c, uo0); });
  static_for<0, TN>([&](auto j_c) { readB(IntC<0>{}, IntC<0>{}, j_c); });

  constexpr int TS = TN / 2 > 0 ? TN / 2 : 1, NTAIL = TN - TS;
  constexpr int NPB = NSLOT / 2 < NTAIL ? NSLOT / 2 : NTAIL, NPA = NSLOT - NPB, PPT = (NPA + TN - 1) / TN;
  static_assert(NPB == BP_NPB(BN) && NSLOT == BP_NSLOT(BN), "prologue and loop agree on the piece schedule");
  int s = 0;
  auto body = [&](auto st_c) {
    constexpr int ST = decltype(st_c)::value, ST1 = (ST + 1) % NS, ST2 = (ST + 2) % NS;
    const int un = 2 * (s + 1) < 2 * nsteps ? 2 * (s + 1) : 0;                // next step's units (clamped: the last fragments are never used)
    const int nu0 = sU[un], nu1 = sU[un + 1];
    // ---- part A: slice 0 (+ the reads of slice 1, + the remaining pieces of group s + 2), then slice-1 tiles [0, TS)
    static_for<0, TN>([&](auto j_c) {
      constexpr int j = decltype(j_c)::value;
      constexpr int a_lo = (j * TM) / TN, a_hi = ((j + 1) * TM) / TN;
      readB(st_c, IntC<1>{}, j_c);
      static_for<a_lo, a_hi>([&](auto i_c) { readA(IntC<1>{}, i_c, uo1); });
      mma_tile(IntC<0>{}, j_c);
      constexpr int p_lo = NPB + j * PPT < NSLOT ? NPB + j * PPT : NSLOT;
      constexpr int p_hi = NPB + (j + 1) * PPT < NSLOT ? NPB + (j + 1) * PPT : NSLOT;
      static_for<p_lo, p_hi>([&](auto k_c) { piece(k_c, s + 2, ST2); });
    });
    static_for<0, TS>([&](auto j_c) { mma_tile(IntC<1>{}, j_c); });
    // ---- the step's barrier: this wave's pieces of group s + 1 have landed; every read of stage ST has been issued and waited for
    __builtin_amdgcn_sched_barrier(0);
#ifdef TDG_STAMPS
    unsigned long long wt1, wt2, wt3;
    TDG_STAMP(wt1);
#endif
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSLOT) : "memory");
#ifdef TDG_STAMPS
    TDG_STAMP(wt2);
#endif
    __syncthreads();
#ifdef TDG_STAMPS
    TDG_STAMP(wt3);
    bw_wait += wt2 - wt1; bw_bar += wt3 - wt2;
#endif
    __builtin_amdgcn_sched_barrier(0);
    // ---- part B: the rest of slice 1; underneath, the slice-0 fragments of step s + 1 and the first pieces of group s + 3
    static_for<0, NTAIL>([&](auto u_c) {
      constexpr int u = decltype(u_c)::value, j = TS + u;
      constexpr int b_lo = (u * TN) / NTAIL, b_hi = ((u + 1) * TN) / NTAIL;
      if constexpr (u == 0) static_for<0, TM>([&](auto i_c) { readA(IntC<0>{}, i_c, nu0); });
      static_for<b_lo, b_hi>([&](auto jj_c) { readB(IntC<ST1>{}, IntC<0>{}, jj_c); });
      mma_tile(IntC<1>{}, IntC<j>{});
      if constexpr (u < NPB) piece(u_c, s + 3, ST);
    });
    uo0 = nu0; uo1 = nu1;
    ++s;
  };
  const int nsteps3 = (nsteps + NS - 1) / NS * NS;          // (whole ring cycles: up to two extra steps on zero-fill pieces)
  while (s < nsteps3) {
    body(IntC<0>{});
    body(IntC<1>{});
    body(IntC<2>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                          // every wave has read its last fragments, every piece has landed: the epilogue stages over the ring
  TDG_STAMP(bs2);

  static_assert(BM * (BN * 2 + 16) + BM * 8 + 2 * (NTHR / BN) * BN * 4 <= OFF_TAB, "epilogue staging must not reach the tables");
  staged_epilogue_bf16<BM, BN, TM, TN, NTHR>(acc, smem, sBias, true, wm * 64, wn * TN, TN, tid, r16, q, m0, n0, M, tile_m, args, cl);
#ifdef TDG_STAMPS
  if (args.stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TDG_STAMP(bs3);
    if (lane == 0) {
      unsigned long long* o = args.stamps + ((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * 8 + wave) * 4;
      o[0] = bs0; o[1] = bs1; o[2] = bs2; o[3] = bs3;
      unsigned long long* o2 = args.stamps + 262144 + ((size_t)(blockIdx.z * gridDim.x + blockIdx.x) * 8 + wave) * 4;
      o2[0] = bw_wait; o2[1] = bw_bar;
    }
  }
#endif
}

// Second half of a split-K forward-type GEMM: out[pixel(m)][n] = epilogue(sum over splits of slab[split][class][m][n]).
// One thread per (class, row, 4 columns); splits are added in ascending order (deterministic).
template <typename T>
__global__ void __launch_bounds__(256) splitk_finish_kernel(const IgArgs args) {
  const int N = args.N, N4 = (N + 3) >> 2;
  const int cls = blockIdx.z;
  const IgClass& cl = args.cls[cls];
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int m = (int)(idx / N4), n = (int)(idx - (long long)m * N4) * 4;
  if (m >= cl.M) return;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int ks = 0; ks < args.ksplit; ++ks) {
    const float* sl = args.slab + ((size_t)(ks * args.nclasses + cls) * args.slab_rows + m) * N + n;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (n + e < N) v[e] += sl[e];
  }
  const unsigned nb = fd_div((unsigned)m, cl.fd_ghw);
  const unsigned rem = (unsigned)m - nb * (unsigned)(cl.GH * cl.GW);
  const unsigned a = fd_div(rem, cl.fd_gw);
  const unsigned b = rem - a * (unsigned)cl.GW;
  const size_t p = ((size_t)(nb * (unsigned)args.OH + a * (unsigned)args.os + (unsigned)cl.oh0) * (unsigned)args.OW +
                    b * (unsigned)args.os + (unsigned)cl.ow0) * (size_t)args.Cso;
  T* out = static_cast<T*>(args.out);
  const T* msk = static_cast<const T*>(args.mask_src);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (n + e >= N) continue;
    float x = apply_act(v[e] + (args.bias ? args.bias[n + e] : 0.f), args.act, args.leak);
    if (args.accumulate) x += to_f32<T>(out[p + n + e]);
    if (args.mask_mode != TDG_MASK_NONE) x *= mask_factor(to_f32<T>(msk[p + n + e]), args.mask_mode, args.leak);
    out[p + n + e] = from_f32<T>(x);
  }
}

// ============================================================================================
// filter gradient: slabs[z][(t,c)][n] = sum over this split's rows of gather(A)[m][(t,c)] * G[m][n]
// ============================================================================================
template <typename T>
struct WgGeom;
template <>
struct WgGeom<bf16_t> {
  static constexpr int MR = 64;   // reduction rows per step (2 x k32)
  static __host__ __device__ constexpr int pitch(int cols) { return ((cols * 2 + 255) / 256) * 256 + 32; }
};
template <>
struct WgGeom<float> {
  static constexpr int MR = 32;   // 8 x k4
  static __host__ __device__ constexpr int pitch(int cols) { return (cols * 4) % 128 == 64 ? cols * 4 : cols * 4 + 64; }
};

// byte offset inside an LDS slab row for the transposed-read layouts
template <typename T>
__device__ __forceinline__ int slab_off(int row, int colbyte, int pitch) {
  if constexpr (sizeof(T) == 2) return row * pitch + (colbyte ^ (((row >> 3) & 1) << 7));
  return row * pitch + colbyte;
}

template <typename T, int BKK, int BN, int WGK, int WGN, bool VECA>
__global__ void __launch_bounds__(256, 2) igemm_wgrad_kernel(const WgArgs args) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int MR = WgGeom<T>::MR;
  constexpr int PA = WgGeom<T>::pitch(BKK), PG = WgGeom<T>::pitch(BN);
  constexpr int WK = BKK / WGK, WN = BN / WGN;
  constexpr int TK = WK / 16, TN = WN / 16;
  constexpr int AVR = BKK / VEC;            // A vectors per slab row
  constexpr int GVR = BN / VEC;             // G vectors per slab row
  constexpr int NAV = MR * AVR / 256;       // A vectors per thread per step
  constexpr int NGV = (MR * GVR + 255) / 256;
  static_assert(WGK * WGN == 4 && WK % 16 == 0 && WN % 16 == 0 && (MR * AVR) % 256 == 0 && 256 % AVR == 0, "tile config");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sG = smem + MR * PA;
  int* sTap = reinterpret_cast<int*>(smem + MR * PA + MR * PG);

  const int tid = threadIdx.x;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_k = bid / args.ntiles_n;
  const int tile_n = bid - tile_k * args.ntiles_n;
  const int kk0 = tile_k * BKK, n0 = tile_n * BN;
  const int split = blockIdx.z;
  const int m_begin = split * args.m_per_split;
  const int m_end = min(args.M, m_begin + args.m_per_split);
  const int SH = args.SH, SW = args.SW, Cs = args.Cs;

  if (tid < IG_MAX_TAPS) sTap[tid] = args.tap[tid];
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rA = make_rsrc(args.src, args.src_bytes);
  const __amdgpu_buffer_rsrc_t rG = make_rsrc(args.g, args.g_bytes);

  // ---- A gather: this thread's kk column is fixed for the whole kernel -----------------------
  const int avc = tid % AVR;                 // vector column
  const int arow0 = tid / AVR;               // first row; rows step by 256/AVR
  constexpr int ARS = 256 / AVR;
  int a_dh = 0, a_dw = 0, a_koff = 0;
  bool a_kok = false;
  if constexpr (VECA) {
    const int kv = (kk0 / VEC) + avc;
    const int CV = (int)args.fd_c.d;
    const int tap = (int)fd_div((unsigned)kv, args.fd_c);
    const int cv = kv - tap * CV;
    a_kok = tap < args.ntaps;
    const int pk = sTap[a_kok ? tap : 0];
    a_dh = tap_dh(pk);
    a_dw = tap_dw(pk);
    a_koff = (a_dh * SW + a_dw) * Cs + cv * VEC;
  }

  i32x4 ra[NAV], rg[NGV];

  auto load_slabs = [&](int mstep) {
    if constexpr (VECA) {
#pragma unroll
      for (int i = 0; i < NAV; ++i) {
        const int m = mstep + arow0 + ARS * i;
        const bool okm = m < m_end;
        const unsigned mm = okm ? (unsigned)m : 0u;
        const unsigned nb = fd_div(mm, args.fd_ghw);
        const unsigned rem = mm - nb * (unsigned)(args.GH * args.GW);
        const unsigned a = fd_div(rem, args.fd_gw);
        const unsigned b = rem - a * (unsigned)args.GW;
        const int ih = (int)a * args.sigma + a_dh, iw = (int)b * args.sigma + a_dw;
        const bool ok = okm && a_kok && (unsigned)ih < (unsigned)SH && (unsigned)iw < (unsigned)SW;
        const unsigned base = ((nb * (unsigned)SH + a * (unsigned)args.sigma) * (unsigned)SW + b * (unsigned)args.sigma) * (unsigned)Cs;
        const unsigned off = ok ? (base + (unsigned)a_koff) * (unsigned)sizeof(T) : OOB_OFFSET;
        ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rA, off, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NGV; ++i) {
      const int v = tid + 256 * i;
      const int row = v / GVR, col = v - row * GVR;
      const int m = mstep + row;
      const int n = n0 + col * VEC;
      const bool ok = (v < MR * GVR) && (m < m_end) && (n < args.N);
      const unsigned off = ok ? ((unsigned)m * (unsigned)args.Gs + (unsigned)n) * (unsigned)sizeof(T) : OOB_OFFSET;
      rg[i] = __builtin_amdgcn_raw_buffer_load_b128(rG, off, 0, 0);
    }
  };

  auto store_slabs = [&]() {
    if constexpr (VECA) {
#pragma unroll
      for (int i = 0; i < NAV; ++i)
        *reinterpret_cast<i32x4*>(sA + slab_off<T>(arow0 + ARS * i, avc * 16, PA)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NGV; ++i) {
      const int v = tid + 256 * i;
      const int row = v / GVR, col = v - row * GVR;
      if (v < MR * GVR) *reinterpret_cast<i32x4*>(sG + slab_off<T>(row, col * 16, PG)) = rg[i];
    }
  };

  auto gather_scalar = [&](int mstep) {
    constexpr int RPT = MR * BKK / 256;
    constexpr int RSTEP = 256 / BKK > 0 ? 256 / BKK : 1;
    static_assert(256 % BKK == 0 || BKK % 256 == 0, "scalar gather mapping");
    const int col = tid % BKK;
    const int kk = kk0 + col;
    const bool k_ok = kk < args.KK;
    const int CV = (int)args.fd_c.d;
    const unsigned tap = fd_div((unsigned)(k_ok ? kk : 0), args.fd_c);
    const int c = (k_ok ? kk : 0) - (int)tap * CV;
    const int pk = sTap[tap];
    const int dh = tap_dh(pk), dw = tap_dw(pk);
#pragma unroll 4
    for (int i = 0; i < RPT; ++i) {
      const int row = tid / BKK + RSTEP * i;
      const int m = mstep + row;
      const bool okm = m < m_end;
      const unsigned mm = okm ? (unsigned)m : 0u;
      const unsigned nb = fd_div(mm, args.fd_ghw);
      const unsigned rem = mm - nb * (unsigned)(args.GH * args.GW);
      const unsigned a = fd_div(rem, args.fd_gw);
      const unsigned b = rem - a * (unsigned)args.GW;
      const int ih = (int)a * args.sigma + dh, iw = (int)b * args.sigma + dw;
      const bool ok = okm && k_ok && (unsigned)ih < (unsigned)SH && (unsigned)iw < (unsigned)SW;
      const unsigned off =
          ok ? (((nb * (unsigned)SH + (unsigned)ih) * (unsigned)SW + (unsigned)iw) * (unsigned)Cs + (unsigned)c) *
                   (unsigned)sizeof(T)
             : OOB_OFFSET;
      const T v = buffer_load_elem<T>(rA, off);
      *reinterpret_cast<T*>(sA + slab_off<T>(row, col * (int)sizeof(T), PA)) = v;
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wk = wave / WGN, wn = wave - wk * WGN;
  const int r16 = lane & 15, q = lane >> 4;

  f32x4 acc[TK][TN];
#pragma unroll
  for (int i = 0; i < TK; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  using Frag = typename Mma<T>::frag;
  load_slabs(m_begin);
  for (int mstep = m_begin; mstep < m_end; mstep += MR) {
    store_slabs();
    if constexpr (!VECA) gather_scalar(mstep);
    __syncthreads();
    if (mstep + MR < m_end) load_slabs(mstep + MR);
    if constexpr (sizeof(T) == 2) {
      // transposed fragments: lane (r16, q) wants slab[ks*32 + 8q + j][col0 + r16], j = 0..7
#pragma unroll
      for (int ks = 0; ks < MR / 32; ++ks) {
        const int rrow = ks * 32 + 8 * q + (r16 >> 2);
        const int rcol = (r16 & 3) * 4;
        Frag fa[TK];
#pragma unroll
        for (int i = 0; i < TK; ++i) {
          const int colb = (wk * WK + i * 16 + rcol) * 2;
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4*)(sA + slab_off<T>(rrow, colb, PA)));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4*)(sA + slab_off<T>(rrow + 4, colb, PA)));
          fa[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int colb = (wn * WN + j * 16 + rcol) * 2;
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4*)(sG + slab_off<T>(rrow, colb, PG)));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4*)(sG + slab_off<T>(rrow + 4, colb, PG)));
          const Frag fg = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
          for (int i = 0; i < TK; ++i) Mma<T>::run(acc[i][j], fg, fa[i]);
        }
      }
    } else {
      // f32: one k4 MFMA per 4 slab rows; lane (r16, q) reads slab[4e + q][col0 + r16]
#pragma unroll
      for (int e = 0; e < MR / 4; ++e) {
        const int rrow = 4 * e + q;
        float fa[TK];
#pragma unroll
        for (int i = 0; i < TK; ++i)
          fa[i] = *reinterpret_cast<const float*>(sA + rrow * PA + (wk * WK + i * 16 + r16) * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float fg = *reinterpret_cast<const float*>(sG + rrow * PG + (wn * WN + j * 16 + r16) * 4);
#pragma unroll
          for (int i = 0; i < TK; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fg, fa[i], acc[i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: lane owns filter row kk (r16) x 4 consecutive n ----------------------------------
  float* slab = args.slabs + (size_t)split * (size_t)args.slab_stride;
  const int CV = (int)args.fd_c.d;
  const int Ceff = VECA ? CV * VEC : CV;
#pragma unroll
  for (int i = 0; i < TK; ++i) {
    const int kk = kk0 + wk * WK + i * 16 + r16;
    if (kk >= args.KK) continue;
    const int tap = kk / Ceff;
    const int c = kk - tap * Ceff;
    if (c >= args.Clog) continue;
    float* rowp = slab + ((size_t)tap * args.Clog + c) * (size_t)args.Nlog;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WN + j * 16 + q * 4;
      if (n + 3 < args.Nlog && (args.Nlog & 3) == 0) {
        *reinterpret_cast<f32x4*>(rowp + n) = acc[i][j];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < args.Nlog) rowp[n + e] = acc[i][j][e];
      }
    }
  }
}

// ============================================================================================
// Large filter gradients (bf16): 256 (filter rows) x 208 (columns) tile, 8 waves as 4 x 2, both
// operand slabs (64 pixel rows per step) streamed by LDS-DMA into a 2-stage ring, one barrier per
// step.  Slab rows are 512 bytes (256 kk / 224+32 n); chunk c of row r sits at physical chunk
// c ^ 2*f(r), f(r) = (r & 3) | (((r >> 3) & 1) << 2): conflict-free for the transposing
// ds_read_b64_tr_b16 fragment reads, and -- with DMA instructions dealt to waves by bits 0 and 2 of
// their index -- a per-lane constant, so each lane's (tap, channel) / column never changes.
// ============================================================================================
#define WD_MR 64
#define WD_ROWB 512
__device__ __forceinline__ int wd_f(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

// Loader of igemm_wgrad_dma_kernel as 8 single wave-instructions ("pieces") per wave and step: piece 2j is the
// j-th 2-row instruction of the gathered operand, piece 2j+1 the same rows of the dense one.
//
// Address arithmetic is kept OFF the vector ALU: the two waves of a SIMD share its vector issue with the MFMAs, and
// the per-piece row -> (image, y, x) decomposition (two fast divisions, six integer multiplies: quarter-rate
// v_mul_lo/hi_u32) cost ~130 VALU instructions per wave and step -- more issue time than the step's MFMAs.  Instead a
// lane's byte offset inside a step is a CONSTANT per piece (computed once), and the step is selected by moving the
// buffer descriptor: base += step * bytes-per-step, num_records = end - base (scalar ALU only).  Rows past the split's
// end, padding taps and padding columns are out-of-range offsets (zero fill) as before.  The dense operand always
// works this way (rows are contiguous); the gathered one when a step covers whole images (64 % (GH*GW) == 0: every
// layer of the 32x32 / 64x64 GAN), FAST = true.  Otherwise (pix2pix, VAE: images larger than a step) the gathered
// operand keeps the per-piece decomposition.
// A buffer window [p, p + rem) that moves by a constant per step; scalar registers only.  (Tensors are < 4 GB -- the C
// ABI's byte counts are 32-bit -- so a non-negative rem fits num_records.)
struct WdCursor {
  unsigned long long p;
  long long rem;
  __device__ __forceinline__ void advance(long long step) { p += (unsigned long long)step; rem -= step; }
  __device__ __forceinline__ i32x4 desc() const {
    const unsigned r = (int)(rem >> 32) < 0 ? 0u : (unsigned)rem;
    return i32x4{(int)(unsigned)p, (int)(unsigned)(p >> 32), (int)r, 0x00020000};
  }
};
// MODE 0: per-piece decomposition of the gathered operand (any geometry).  MODE 1 ("FAST"): a step covers whole images
// (64 % (GH*GW) == 0): constant lane offsets, moving descriptor.  MODE 2 ("RECT"): a step covers a rectangle of ONE image
// -- 64 / GW whole rows (64 % GW == 0) or a 64-column piece of one row (GW % 64 == 0), images being multiples of 64 rows --
// so a row's (y, x) is a lane constant plus the step's scalar corner (sa, sb): the descriptor moves to the corner
// (minus a halo, so that taps above / left of it keep non-negative offsets) and the only per-piece vector work left is
// the bounds test of the tap against the image (two adds, two compares, one select) instead of two divisions and six
// multiplies.  pix2pix / VAE layers and the critic's c1.
struct WdDescs {
  i32x4 a1, a2, g;
  int sa, sb;                                // MODE 2: sigma * (first row, first column) of the step's rectangle
};
#define WD_HALO 8
template <int MODE>
struct WdLoader {
  static constexpr bool FAST = MODE == 1;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  unsigned long long baseA, baseA2, baseG;   // baseA2 = src2 - img_switch images: rows >= m_switch index it like baseA
  long long endA, endA2, endG;               // bytes from the base that rows < min(m_end, m_switch) / < m_end may touch
  long long stepA, stepG;                    // bytes per 64-row step (FAST: whole images)
  unsigned voA[4], voG[4];                   // per-lane byte offsets inside a step (OOB_OFFSET: padding)
  int ihl[4], iwl[4];                        // MODE 2: lane part of the tap's source row / column (huge: padding chunk)
  i32x4 rA, rA2;                             // MODE 0: whole-tensor descriptors
  unsigned long long src1, src2p;            // MODE 2: tensor bases and sizes
  long long src1_bytes, src2_bytes;
  unsigned lds0;                             // LDS byte address of the ring
  int m_begin, m_switch, img_switch;         // rows >= m_switch gather from the second tensor (its image 0 = image img_switch)
  FastDiv fd_ghw, fd_gw;
  int GHW, GW, GH, SH, SW, Cs, sigma, m_end;
  int a_dh, a_dw, a_koff, a_kok, hrow;
  int Ibase;                        // b0 | b2 << 2 | b4 << 4
  // cursors of the three operands at the split's first step / their descriptors / one step forward
  struct Cursors {
    WdCursor a1, a2, g;
    int nb, a0, b0;                 // MODE 2: image and corner (grid units) of the step
  };
  __device__ __forceinline__ Cursors begin() const {
    Cursors c;
    c.a1.p = baseA; c.a1.rem = endA;
    c.a2.p = baseA2; c.a2.rem = endA2;
    c.g.p = baseG; c.g.rem = endG;
    c.nb = m_begin / GHW;
    const int rem = m_begin - c.nb * GHW;
    c.a0 = rem / GW;
    c.b0 = rem - c.a0 * GW;
    return c;
  }
  __device__ __forceinline__ WdDescs descs(const Cursors& c) const {
    WdDescs d;
    d.g = c.g.desc();
    d.sa = d.sb = 0;
    if constexpr (MODE == 1) {
      d.a1 = c.a1.desc();
      d.a2 = c.a2.desc();
    } else if constexpr (MODE == 2) {
      const bool second = c.nb >= img_switch;
      const long long e = ((long long)(second ? c.nb - img_switch : c.nb) * SH * SW + (long long)(c.a0 * sigma) * SW + c.b0 * sigma -
                           (WD_HALO * SW + WD_HALO)) * Cs * 2;                                   // bytes; may be < 0 at the first rows
      WdCursor w;
      w.p = (second ? src2p : src1) + (unsigned long long)e;
      w.rem = (second ? src2_bytes : src1_bytes) - e;
      d.a1 = d.a2 = w.desc();
      d.sa = c.a0 * sigma;
      d.sb = c.b0 * sigma;
    } else {
      d.a1 = rA;
      d.a2 = rA2;
    }
    return d;
  }
  __device__ __forceinline__ void advance(Cursors& c) const {
    c.g.advance(stepG);
    if constexpr (MODE == 1) {
      c.a1.advance(stepA);
      c.a2.advance(stepA);
    } else if constexpr (MODE == 2) {
      c.b0 += GW < WD_MR ? GW : WD_MR;
      if (c.b0 >= GW) {
        c.b0 = 0;
        c.a0 += GW < WD_MR ? WD_MR / GW : 1;
        if (c.a0 >= GH) { c.a0 = 0; c.nb += 1; }
      }
    }
  }
  template <int P>
  __device__ __forceinline__ void piece(const WdDescs& d, int mstep, int stage_off) const {
    constexpr int j = P >> 1;
    constexpr int SLABB = WD_MR * WD_ROWB;
    const int I = Ibase | ((j & 1) << 1) | ((j >> 1) << 3);
    if constexpr ((P & 1) == 0) {
      if constexpr (MODE == 2) {
        const int ih = d.sa + ihl[j], iw = d.sb + iwl[j];
        const int ok = ((unsigned)ih < (unsigned)SH) & ((unsigned)iw < (unsigned)SW);
        lds_dma_b128(d.a1, ok ? voA[j] : OOB_OFFSET, lds0 + (unsigned)(stage_off + I * 1024));
        return;
      }
      // wave-uniform (m_switch is even: both rows on one side); scalar, so the descriptor is picked by s_cselect
      const bool second = __builtin_amdgcn_readfirstlane((int)(mstep + 2 * I >= m_switch)) != 0;
      unsigned offa;
      if constexpr (FAST) {
        offa = voA[j];
      } else {
        const int m = mstep + 2 * I + hrow;
        const int okm = m < m_end;
        const unsigned mm = okm ? (unsigned)m : 0u;
        const unsigned nbg = fd_div(mm, fd_ghw);
        const unsigned rem = mm - nbg * (unsigned)GHW;
        const unsigned nb = second ? nbg - (unsigned)img_switch : nbg;
        const unsigned a = fd_div(rem, fd_gw);
        const unsigned b = rem - a * (unsigned)GW;
        const int ih = (int)a * sigma + a_dh, iw = (int)b * sigma + a_dw;
        const int ok = okm & a_kok & ((unsigned)ih < (unsigned)SH) & ((unsigned)iw < (unsigned)SW);
        const unsigned base = ((nb * (unsigned)SH + a * (unsigned)sigma) * (unsigned)SW + b * (unsigned)sigma) * (unsigned)Cs;
        offa = ok ? (base + (unsigned)a_koff) * 2u : OOB_OFFSET;
      }
      lds_dma_b128(second ? d.a2 : d.a1, offa, lds0 + (unsigned)(stage_off + I * 1024));
    } else {
      lds_dma_b128(d.g, voG[j], lds0 + (unsigned)(stage_off + SLABB + I * 1024));
    }
  }
  template <int P = 0>
  __device__ __forceinline__ void all_pieces(const WdDescs& d, int mstep, int stage_off) const {
    if constexpr (P < 8) {
      piece<P>(d, mstep, stage_off);
      all_pieces<P + 1>(d, mstep, stage_off);
    }
  }
};

// Diagnostic builds only (-DTDG_WG_ABLATE=n, never the product library): 1 = the filter-gradient K loop without its
// MFMAs (fragments still read), 3 = without LDS-DMA pieces, 4 = without fragment reads.  Measured on c3 (1024 images, 0.311 ms): 0.238 ms
// without MFMAs, 0.229 ms without the DMA pieces -- the 168 KB of 8-byte transposing fragment reads per step are the
// floor (~64 B/clk), the DMA issue costs a quarter on top.
#ifndef TDG_WG_ABLATE
#define TDG_WG_ABLATE 0
#endif

template <int BN, int MODE>
__global__ void __launch_bounds__(512, 2) igemm_wgrad_dma_kernel(const WgArgs args) {
  constexpr bool FAST = MODE == 1;
  using T = bf16_t;
  constexpr int BKK = 256, VEC = 8;
  constexpr int TK = 4, TN = (BN / 16 + 1) / 2, TN1 = BN / 16 - TN;
  constexpr int SLAB = WD_MR * WD_ROWB;              // 32 KB per operand per stage
  constexpr int STAGE = 2 * SLAB;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef __attribute__((address_space(3))) bf16x4* lds_b4_t;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* sTap = reinterpret_cast<int*>(smem + 2 * STAGE);

  const int tid = threadIdx.x;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_k = bid / args.ntiles_n;
  const int tile_n = bid - tile_k * args.ntiles_n;
  const int kk0 = tile_k * BKK, n0 = tile_n * BN;
  const int split = blockIdx.z;
  const int m_begin = split * args.m_per_split;
  const int m_end = min(args.M, m_begin + args.m_per_split);
  const int SH = args.SH, SW = args.SW, Cs = args.Cs, sigma = args.sigma, Gs = args.Gs, GHW = args.GH * args.GW, GW = args.GW;

  if (tid < IG_MAX_TAPS) sTap[tid] = args.tap[tid];
  __syncthreads();

  const i32x4 rA = make_rsrc_words(args.src, args.src_bytes);
  const i32x4 rG = make_rsrc_words(args.g, args.g_bytes);

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int hrow = lane >> 5;                                   // row inside a 2-row DMA instruction
  const int b0 = wave & 1, b2 = (wave >> 1) & 1, b4 = (wave >> 2) & 1;
  const int lch = (lane & 31) ^ (2 * ((2 * b0 + hrow) | (b2 << 2)));   // this lane's logical 16-byte chunk (0..31)

  // A: (tap, channel vector) of this lane's chunk -- fixed for the whole kernel
  int a_dh = 0, a_dw = 0, a_koff = 0;
  bool a_kok;
  {
    const int kv = (kk0 >> 3) + lch;
    const int CV = (int)args.fd_c.d;
    const int tap = (int)fd_div((unsigned)kv, args.fd_c);
    const int cv = kv - tap * CV;
    a_kok = tap < args.ntaps;
    const int pk = sTap[a_kok ? tap : 0];
    a_dh = tap_dh(pk);
    a_dw = tap_dw(pk);
    a_koff = (a_dh * SW + a_dw) * Cs + cv * VEC;
  }
  const int g_n = n0 + lch * VEC;
  const bool g_nok = (lch * VEC < TN * 32) && (g_n < args.N);

  WdLoader<MODE> ld;
  {
    const long long imgbytes = (long long)SH * SW * Cs * 2;
    const int m_sw = m_end < args.m_switch ? m_end : args.m_switch;
    ld.rA = rA;
    ld.rA2 = make_rsrc_words(args.src2 ? args.src2 : args.src, args.src2 ? args.src2_bytes : args.src_bytes);
    ld.baseA = (unsigned long long)args.src + (unsigned long long)(m_begin / GHW) * (unsigned long long)imgbytes;
    ld.baseA2 = (unsigned long long)(args.src2 ? args.src2 : args.src) +
                (unsigned long long)((long long)(m_begin / GHW - args.img_switch) * imgbytes);
    ld.endA = (long long)(m_sw / GHW - m_begin / GHW) * imgbytes;
    ld.endA2 = (long long)(m_end / GHW - m_begin / GHW) * imgbytes;
    ld.stepA = (long long)(WD_MR / (GHW > 0 ? GHW : 1)) * imgbytes;
    ld.baseG = (unsigned long long)args.g + (unsigned long long)m_begin * (unsigned long long)(Gs * 2);
    ld.endG = (long long)(m_end - m_begin) * (Gs * 2);
    ld.stepG = (long long)WD_MR * (Gs * 2);
  }
  ld.m_begin = m_begin; ld.m_switch = args.m_switch; ld.img_switch = args.img_switch;
  ld.lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
  ld.fd_ghw = args.fd_ghw; ld.fd_gw = args.fd_gw;
  ld.GHW = GHW; ld.GW = GW; ld.GH = args.GH; ld.SH = SH; ld.SW = SW; ld.Cs = Cs; ld.sigma = sigma; ld.m_end = m_end;
  ld.src1 = (unsigned long long)args.src; ld.src1_bytes = (long long)args.src_bytes;
  ld.src2p = (unsigned long long)(args.src2 ? args.src2 : args.src); ld.src2_bytes = (long long)(args.src2 ? args.src2_bytes : args.src_bytes);
  ld.a_dh = a_dh; ld.a_dw = a_dw; ld.a_koff = a_koff; ld.a_kok = (int)a_kok; ld.hrow = hrow;
  ld.Ibase = b0 | (b2 << 2) | (b4 << 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int I = ld.Ibase | ((j & 1) << 1) | ((j >> 1) << 3);
    const int ml = 2 * I + hrow;                                     // row inside a step
    ld.voG[j] = g_nok ? (unsigned)(ml * Gs + g_n) * 2u : OOB_OFFSET;
    if constexpr (FAST) {
      const unsigned nb = fd_div((unsigned)ml, args.fd_ghw);
      const unsigned rem = (unsigned)ml - nb * (unsigned)GHW;
      const unsigned a = fd_div(rem, args.fd_gw);
      const unsigned b = rem - a * (unsigned)GW;
      const int ih = (int)a * sigma + a_dh, iw = (int)b * sigma + a_dw;
      const bool ok = a_kok && (unsigned)ih < (unsigned)SH && (unsigned)iw < (unsigned)SW;
      const unsigned base = ((nb * (unsigned)SH + a * (unsigned)sigma) * (unsigned)SW + b * (unsigned)sigma) * (unsigned)Cs;
      ld.voA[j] = ok ? (base + (unsigned)a_koff) * 2u : OOB_OFFSET;
    } else if constexpr (MODE == 2) {
      const int a_l = GW < WD_MR ? ml / GW : 0, b_l = GW < WD_MR ? ml - (ml / GW) * GW : ml;
      ld.ihl[j] = a_kok ? a_l * sigma + a_dh : 0x40000000;          // a padding chunk never passes the bounds test
      ld.iwl[j] = b_l * sigma + a_dw;
      ld.voA[j] = (unsigned)(((a_l * sigma + WD_HALO) * SW + b_l * sigma + WD_HALO) * Cs + a_koff) * 2u;
    } else {
      ld.voA[j] = 0;
    }
    if constexpr (MODE != 2) { ld.ihl[j] = 0; ld.iwl[j] = 0; }
  }

  const int r16 = lane & 15, q = lane >> 4;
  const int wk = wave & 3, wn = wave >> 2;
  const int tnw = wn == 0 ? TN : TN1;
  // transposing fragment reads: row = ks*32 + 8q + 4h + (r16>>2); 8 bytes at column col0 + (r16&3)*4
  const int frow = 8 * q + (r16 >> 2);
  const int fsw = 2 * ((r16 >> 2) | ((q & 1) << 2));             // wd_f of that row (the + 4h / + 32 ks terms do not change it)
  const int fcol = (r16 & 3) * 4;

  f32x4 acc[TK][TN];
#pragma unroll
  for (int i = 0; i < TK; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto frag = [&](const char* slab, int ks, int col0) -> bf16x8 {
#if TDG_WG_ABLATE == 4
    bf16x8 z = {};
    asm volatile("" : "+v"(z));
    return z;
#endif
    const int col = col0 + fcol;
    const int cb = (((col >> 3) ^ fsw) << 4) + ((col & 7) << 1);
    const char* p = slab + (ks * 32 + frow) * WD_ROWB + cb;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_t)p);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_t)(p + 4 * WD_ROWB));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };

  // ---- main loop -----------------------------------------------------------------------------------------------
  // A step = 64 pixel rows = two k-slices of 32.  Fragments are double-buffered in registers by slice: while the MFMAs of
  // slice 0 run, every fragment of slice 1 is requested (a whole slice of lookahead instead of one column tile: the
  // transposing reads' latency is off the MFMA chain).  The step's barrier is SKEWED: it sits TS column tiles into
  // slice 1; behind it the fragments of the NEXT step's slice 0 are requested and the remaining MFMAs of the current
  // step (operands already in registers) cover their latency.  The LDS-DMA pieces of step s+2 are dealt one per column
  // tile from that barrier on (the stage they overwrite was last read before it), so the `vmcnt(0)` in front of the
  // next barrier waits for loads issued >= 6 column tiles earlier.  Rows past m_end are out-of-range sources (zero
  // fill), so the loop needs no peeled last step: the trailing pieces and the fragments read behind the last barrier
  // are never used.
  unsigned long long ws0 = 0, ws1 = 0, ws2 = 0, ws3 = 0, w_mma = 0, w_wait = 0, w_bar = 0;
  TDG_STAMP(ws0);
  auto run = [&](auto tnw_c) {
    constexpr int TNW = decltype(tnw_c)::value;        // column tiles of this wave (208 columns: 7 | 6)
    constexpr int TS = TNW >= 6 ? 3 : TNW / 2;         // slice-1 tiles in front of the barrier
    constexpr int NTAIL = TNW - TS;                    // ... and behind it
    constexpr int NPA = 8 - (NTAIL < 4 ? NTAIL : 4);   // pieces issued in part A (the rest ride on the tail tiles)
    constexpr int PPT = (NPA + TNW - 1) / TNW;         // ... per slice-0 column tile (1 at 208 columns, 2 at 128)
    bf16x8 FA[2][TK], FG[2][TNW];
    const int col_g = wn * TN * 16, col_a = wk * 64;
    auto mma_tile = [&](auto ks_c, auto t_c) {
      constexpr int ks = decltype(ks_c)::value, t = decltype(t_c)::value;
#if TDG_WG_ABLATE == 1
      asm volatile("" ::"v"(FG[ks][t]));
#pragma unroll
      for (int i = 0; i < TK; ++i) asm volatile("" ::"v"(FA[ks][i]));
#else
#pragma unroll
      for (int i = 0; i < TK; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FG[ks][t], FA[ks][i], acc[i][t], 0, 0, 0);
#endif
    };
    // prologue: step 0 into stage 0, its slice-0 fragments, and the first pieces of step 1
    WdDescs dn1;
    typename WdLoader<MODE>::Cursors cur = ld.begin();
    ld.all_pieces(ld.descs(cur), m_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the asm-issued LDS-DMA is not counted by the compiler
    __syncthreads();
    {
      const char* sA = smem;
      const char* sG = sA + SLAB;
      static_for<0, TK>([&](auto i_c) { constexpr int i = decltype(i_c)::value; FA[0][i] = frag(sA, 0, col_a + i * 16); });
      static_for<0, TNW>([&](auto t_c) { constexpr int t = decltype(t_c)::value; FG[0][t] = frag(sG, 0, col_g + t * 16); });
      ld.advance(cur);
      dn1 = ld.descs(cur);
      const WdDescs d1 = dn1;
      static_for<0, 8 - NPA>([&](auto p_c) { ld.template piece<decltype(p_c)::value>(d1, m_begin + WD_MR, STAGE); });
    }
    TDG_STAMP(ws1);
    int stage = 0;
    for (int mstep = m_begin; mstep < m_end; mstep += WD_MR) {
      const char* sA = smem + stage * STAGE;
      const char* sG = sA + SLAB;
      const int nstage = (stage ^ 1) * STAGE;
      ld.advance(cur);
      const WdDescs dn2 = ld.descs(cur);                 // step s+2 (dn1: step s+1)
      // ---- part A: slice 0 (+ the reads of slice 1, + the remaining pieces of step s+1), then slice-1 tiles [0, TS)
      static_for<0, TNW>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        FG[1][t] = frag(sG, 1, col_g + t * 16);
        if constexpr (t < TK) FA[1][t] = frag(sA, 1, col_a + t * 16);
        mma_tile(IntC<0>{}, t_c);
        constexpr int p_lo = 8 - NPA + t * PPT < 8 ? 8 - NPA + t * PPT : 8;
        constexpr int p_hi = 8 - NPA + (t + 1) * PPT < 8 ? 8 - NPA + (t + 1) * PPT : 8;
        constexpr int np = TDG_WG_ABLATE != 3 ? p_hi - p_lo : 0;
        static_for<p_lo, p_lo + np>([&](auto p_c) { ld.template piece<decltype(p_c)::value>(dn1, mstep + WD_MR, nstage); });
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * (1 + (t < TK ? 1 : 0)), 0);
        __builtin_amdgcn_sched_group_barrier(0x008, TK, 0);
        if constexpr (np > 0) __builtin_amdgcn_sched_group_barrier(0x020, np, 0);
      });
      static_for<0, TS>([&](auto t_c) {
        mma_tile(IntC<1>{}, t_c);
        __builtin_amdgcn_sched_group_barrier(0x008, TK, 0);
      });
      // ---- the step's barrier: every wave's pieces of step s+1 have landed, every read of this stage has returned
      __builtin_amdgcn_sched_barrier(0);
#ifdef TDG_STAMPS
      unsigned long long wt1, wt2, wt3;
      TDG_STAMP(wt1);
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef TDG_STAMPS
      TDG_STAMP(wt2);
#endif
      __syncthreads();
#ifdef TDG_STAMPS
      TDG_STAMP(wt3);
      w_wait += wt2 - wt1; w_bar += wt3 - wt2;
#endif
      __builtin_amdgcn_sched_barrier(0);
      // ---- part B: the rest of slice 1; underneath, the fragments of the next step's slice 0 and the first pieces of
      // step s+2 (into the stage this step has just finished reading)
      const char* nA = smem + nstage;
      const char* nG = nA + SLAB;
      static_for<0, NTAIL>([&](auto u_c) {
        constexpr int u = decltype(u_c)::value, t = TS + u;
        // reads: first tail tile takes what tile 0 of the next step needs (all of A, column tile 0), the others share the rest
        constexpr int g_lo = u == 0 ? 0 : 1 + ((TNW - 1) * (u - 1)) / (NTAIL - 1 > 0 ? NTAIL - 1 : 1);
        constexpr int g_hi = u == 0 ? 1 : (u == NTAIL - 1 ? TNW : 1 + ((TNW - 1) * u) / (NTAIL - 1 > 0 ? NTAIL - 1 : 1));
        if constexpr (u == 0) static_for<0, TK>([&](auto i_c) { constexpr int i = decltype(i_c)::value; FA[0][i] = frag(nA, 0, col_a + i * 16); });
        static_for<g_lo, g_hi>([&](auto g_c) { constexpr int g = decltype(g_c)::value; FG[0][g] = frag(nG, 0, col_g + g * 16); });
        mma_tile(IntC<1>{}, IntC<t>{});
        constexpr bool has_piece = u < 8 - NPA && TDG_WG_ABLATE != 3;
        if constexpr (has_piece) ld.template piece<u>(dn2, mstep + 2 * WD_MR, stage * STAGE);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * ((u == 0 ? TK : 0) + (g_hi - g_lo)), 0);
        __builtin_amdgcn_sched_group_barrier(0x008, TK, 0);
        if constexpr (has_piece) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      });
      stage ^= 1;
      dn1 = dn2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (out-of-range, zero-fill) pieces
  };
  if constexpr (TN1 == TN) {
    run(IntC<TN>{});
  } else {
    if (wn == 0) run(IntC<TN>{});
    else run(IntC<TN1>{});
  }
  TDG_STAMP(ws2);

  // ---- epilogue: lane owns filter row kk (r16) x 4 consecutive n ----------------------------------
  float* slab = args.slabs + (size_t)split * (size_t)args.slab_stride;
  const int Ceff = (int)args.fd_c.d * VEC;
#pragma unroll
  for (int i = 0; i < TK; ++i) {
    const int kk = kk0 + wk * 64 + i * 16 + r16;
    if (kk >= args.KK) continue;
    const int tap = kk / Ceff;
    const int c = kk - tap * Ceff;
    if (c >= args.Clog) continue;
    float* rowp = slab + ((size_t)tap * args.Clog + c) * (size_t)args.Nlog;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 16 + q * 4;
      if (j >= tnw) continue;
      if (n + 3 < args.Nlog && (args.Nlog & 3) == 0) {
        *reinterpret_cast<f32x4*>(rowp + n) = acc[i][j];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < args.Nlog) rowp[n + e] = acc[i][j][e];
      }
    }
  }
#ifdef TDG_STAMPS
  if (args.stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TDG_STAMP(ws3);
    if (lane == 0) {
      unsigned long long* o = args.stamps + (((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8 + wave) * 8;
      o[0] = ws0; o[1] = ws1; o[2] = ws2; o[3] = ws3; o[4] = (ws2 - ws1) - w_wait - w_bar; o[5] = w_wait; o[6] = w_bar;
    }
  }
#endif
}

// dw[i] = beta*dw[i] + sum_z slabs[z][i]   (fixed summation order: 16 z-lanes, then lane order)
__global__ void __launch_bounds__(256) slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                         size_t n, int nsplit, size_t stride, float beta) {
  __shared__ f32x4 sh[16][16];
  const int tx = threadIdx.x & 15, tz = threadIdx.x >> 4;
  const size_t n4 = (n + 3) >> 2;                      // slabs are padded to a multiple of 4 floats
  const size_t i = (size_t)blockIdx.x * 16 + tx;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < n4)
    for (int z = tz; z < nsplit; z += 16) s += *reinterpret_cast<const f32x4*>(slabs + (size_t)z * stride + 4 * i);
  sh[tz][tx] = s;
  __syncthreads();
  if (tz != 0 || i >= n4) return;
  s = sh[0][tx];
#pragma unroll
  for (int k = 1; k < 16; ++k) s += sh[k][tx];
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (4 * i + e < n) dw[4 * i + e] = (beta != 0.f ? beta * dw[4 * i + e] : 0.f) + s[e];
}

// few slabs (<= 8): one 16-byte column per thread, every slab's load in flight before the sum (z ascending: the same
// fixed order).  The 16-z-lane kernel above leaves 13 of 16 lanes idle at 3 slabs (1.8 TB/s).
__global__ void __launch_bounds__(256) slab_reduce_few_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                             size_t n, int nsplit, size_t stride, float beta) {
  const size_t n4 = (n + 3) >> 2;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 v[8];
#pragma unroll
  for (int z = 0; z < 8; ++z)
    v[z] = z < nsplit ? *reinterpret_cast<const f32x4*>(slabs + (size_t)z * stride + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 s = v[0];
#pragma unroll
  for (int z = 1; z < 8; ++z) s += v[z];
  if (4 * i + 3 < n) {
    f32x4 o = s;
    if (beta != 0.f) o += beta * *reinterpret_cast<const f32x4*>(dw + 4 * i);
    *reinterpret_cast<f32x4*>(dw + 4 * i) = o;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * i + e < n) dw[4 * i + e] = (beta != 0.f ? beta * dw[4 * i + e] : 0.f) + s[e];
  }
}

// ============================================================================================
// filter packing: f32 master [taps][Cdim][Kdim] -> packed [rows][Kp] (dtype T), zero padded
//   element (row r, tap index ti, channel c) = w[tap_ids[ti] * stride_tap + r*stride_row + c*stride_ch]
// ============================================================================================
struct PackArgs {
  const float* w;
  void* out;
  int rows, ntaps, C, Ceff, Kp;
  int CK;                // channels of a tap per K slice (= Ceff: plain (tap, channel) order; IgArgs.fd_ck otherwise)
  int stride_tap, stride_row, stride_ch;
  FastDiv fd_slice, fd_ck;   // / (ntaps * CK), / CK   (pack_finish)
  unsigned char tap_ids[IG_MAX_TAPS];
};
static inline void pack_finish(PackArgs* a) {
  a->fd_slice = make_fastdiv((uint32_t)(a->ntaps * a->CK));
  a->fd_ck = make_fastdiv((uint32_t)a->CK);
}

// PACK_RT (row tile of 32, k tile of PACK_TK = 64) tiles per workgroup, each with an LDS transpose so both sides coalesce; 8
// elements per thread and tile, two consecutive k per store (the 32 x 32 tile with 2-byte stores ran at 1.7 TB/s).  The K index ->
// (slice, tap, channel) -> master offset of a thread's k's does not depend on the row tile: it is computed once per workgroup
// (round 4: one tile per workgroup with two runtime divisions per element packed the critic's 10 M parameters in 38 us =
// 2.1 TB/s -- 10 127 workgroups of 8 elements per thread).
#define PACK_TK 64
#define PACK_RT 4
template <typename T>
__device__ __forceinline__ void pack_tiles(const PackArgs& a, int bx, int byg, float (*tile)[33]) {
  const int k0 = bx * PACK_TK;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const bool row_fast = a.stride_row == 1;                  // which source index is contiguous
  const int kmax = a.ntaps * a.Ceff;
  // master offset of K index k without its row term; -1: a zero (K padding, channel padding)
  auto k_src = [&](int k) -> int {
    if (k >= kmax) return -1;
    const int sl = (int)fd_div((unsigned)k, a.fd_slice), kr = k - sl * (a.ntaps * a.CK);
    const int ti = (int)fd_div((unsigned)kr, a.fd_ck), c = sl * a.CK + (kr - ti * a.CK);
    return c < a.C ? (int)a.tap_ids[ti] * a.stride_tap + c * a.stride_ch : -1;
  };
  int ks[PACK_TK / 8];
  if (row_fast) {                                           // lanes walk rows; k = k0 + ty + 8p
#pragma unroll
    for (int p = 0; p < PACK_TK / 8; ++p) ks[p] = k_src(k0 + ty + 8 * p);
  } else {                                                  // lanes walk k (two halves of the k tile); rows = r0 + ty + 8p
#pragma unroll
    for (int h = 0; h < PACK_TK / 32; ++h) ks[h] = k_src(k0 + h * 32 + tx);
  }
  T* out = static_cast<T*>(a.out);
  for (int s = 0; s < PACK_RT; ++s) {
    const int r0 = (byg * PACK_RT + s) * 32;
    if (r0 >= a.rows) break;                                // (uniform)
    if (row_fast) {
      const int r = r0 + tx;
#pragma unroll
      for (int p = 0; p < PACK_TK / 8; ++p)
        tile[ty + 8 * p][tx] = (ks[p] >= 0 && r < a.rows) ? a.w[(size_t)ks[p] + (size_t)r] : 0.f;     // tile[k_local][r_local]
    } else {
#pragma unroll
      for (int h = 0; h < PACK_TK / 32; ++h)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int i = ty + 8 * p, r = r0 + i;
          tile[h * 32 + tx][i] = (ks[h] >= 0 && r < a.rows) ? a.w[(size_t)ks[h] + (size_t)r * a.stride_row] : 0.f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int rl = ty + 8 * p;
      const int r = r0 + rl, k = k0 + 2 * tx;
      if (r >= a.rows || k >= a.Kp) continue;
      const float v0 = tile[2 * tx][rl], v1 = tile[2 * tx + 1][rl];
      if (k + 1 < a.Kp && ((a.Kp & 1) == 0)) {                // (row pitch even: the pair is aligned)
        if constexpr (sizeof(T) == 2) {
          typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));
          *reinterpret_cast<bf16x2_t*>(out + (size_t)r * a.Kp + k) = bf16x2_t{(bf16_t)v0, (bf16_t)v1};
        } else {
          *reinterpret_cast<float2*>(out + (size_t)r * a.Kp + k) = float2{v0, v1};
        }
      } else {
        out[(size_t)r * a.Kp + k] = from_f32<T>(v0);
        if (k + 1 < a.Kp) out[(size_t)r * a.Kp + k + 1] = from_f32<T>(v1);
      }
    }
    __syncthreads();
  }
}

template <typename T>
__global__ void __launch_bounds__(256) pack_filter_kernel(const PackArgs a) {
  __shared__ float tile[PACK_TK][33];
  pack_tiles<T>(a, blockIdx.x, blockIdx.y, tile);
}

// merged filter of bwd_fused_kernel from the f32 master [kh][kw][c][k]
struct FusedPackArgs {
  const float* w;
  bf16_t* out;
  int C, K, KH, KW, pad_t, pad_l;
  int nhm, nwm, dh_min, dw_min, KP, wpitch;
};
__device__ __forceinline__ void pack_fused_elem(const FusedPackArgs& a, int i) {
  if (i >= 16 * a.wpitch) return;
  const int col = i / a.wpitch, r = i - col * a.wpitch;
  float v = 0.f;
  const int cls = col / a.C, c = col - cls * a.C;
  const int tap = r / a.KP, k = r - tap * a.KP;
  if (cls < 4 && tap < a.nhm * a.nwm && k < a.K) {
    const int th = tap / a.nwm, tw = tap - th * a.nwm;
    const int kh = (cls >> 1) + a.pad_t - 2 * (a.dh_min + th), kw = (cls & 1) + a.pad_l - 2 * (a.dw_min + tw);
    if (kh >= 0 && kh < a.KH && kw >= 0 && kw < a.KW) v = a.w[(((size_t)kh * a.KW + kw) * a.C + c) * a.K + k];
  }
  a.out[i] = (bf16_t)v;
}
__global__ void __launch_bounds__(256) pack_fused_kernel(const FusedPackArgs a) { pack_fused_elem(a, blockIdx.x * 256 + threadIdx.x); }

// filter of thin_fwd_kernel from the f32 master [kh][kw][c][k]: rows n (zero beyond N), k = tap * C + c
struct ThinPackArgs {
  const float* w;
  bf16_t* out;
  int C, N, taps, Kp, WP, rows;
  int tile_rows, tile_cols;   // filter rows staged per column tile / output columns of a tile (224 / 208 or 64 / 64)
};
__device__ __forceinline__ void pack_thin_elem(const ThinPackArgs& a, int i) {
  if (i >= a.rows * a.WP) return;
  const int r = i / a.WP, k = i - r * a.WP;
  const int n = (r / a.tile_rows) * a.tile_cols + (r % a.tile_rows);     // row r of tile t = r / tile_rows is output column t * tile_cols + r % tile_rows
  float v = 0.f;
  if ((r % a.tile_rows) < a.tile_cols && n < a.N && k < a.taps * a.C) {
    const int tap = k / a.C, c = k - tap * a.C;
    v = a.w[((size_t)tap * a.C + c) * a.N + n];
  }
  a.out[i] = (bf16_t)v;
}
__global__ void __launch_bounds__(256) pack_thin_kernel(const ThinPackArgs a) { pack_thin_elem(a, blockIdx.x * 256 + threadIdx.x); }

// several packing jobs in one launch: block b belongs to the job j with start[j] <= b < start[j+1]
#define PACK_MULTI_MAX 32
// ... behind them thin_blocks blocks of the thin-input filter and fused_blocks blocks of the fused-class backward-data filter of the same
// network (one element per thread; they were two launches of ~5 us each per packing call)
struct PackMultiArgs {
  int njobs;
  int start[PACK_MULTI_MAX + 1];
  int thin_blocks, fused_blocks;
  ThinPackArgs thin;
  FusedPackArgs fused;
  PackArgs job[PACK_MULTI_MAX];
};
static_assert(sizeof(PackMultiArgs) <= 4096, "kernel argument block");

template <typename T>
__global__ void __launch_bounds__(256) pack_multi_kernel(const PackMultiArgs m) {
  __shared__ float tile[PACK_TK][33];
  const int ntile = m.start[m.njobs];
  if ((int)blockIdx.x >= ntile) {
    const int b = blockIdx.x - ntile;
    if (b < m.thin_blocks) pack_thin_elem(m.thin, b * 256 + threadIdx.x);
    else pack_fused_elem(m.fused, (b - m.thin_blocks) * 256 + threadIdx.x);
    return;
  }
  int j = 0;
  while (j + 1 < m.njobs && (int)blockIdx.x >= m.start[j + 1]) ++j;
  const PackArgs& a = m.job[j];
  const int local = blockIdx.x - m.start[j];
  const int gx = (a.Kp + PACK_TK - 1) / PACK_TK;
  const int by = local / gx;
  pack_tiles<T>(a, local - by * gx, by, tile);
}

// ============================================================================================
// Stride-2 backward-data with a thin big side (4*C <= 16 columns: the discriminator's first layer and the
// generator's last): all four output-parity classes in ONE pass.  A workgroup stages the small-side rows
// it needs (TA anchor rows + halo, every channel) in LDS once and multiplies them with the merged filter
// W[(class, c)][tap of the union tap grid][k] (zeros where a class has no such tap), 16 columns wide.
// The class-per-launch-z form re-read the small side once per class and tap through L2 (0.78 ms for
// 1536 x 16x16x200 -> 32x32x3); here it is read once from HBM.   bf16 only.
// ============================================================================================
struct FusedBwdArgs {
  const bf16_t* y;       // small side [n][SH][SW][Cs]
  const bf16_t* w;       // merged filter [16][wpitch]
  bf16_t* x;             // big side [n][OH][OW][Cso]
  const float* bias;
  const bf16_t* mask_src;
  int SH, SW, Cs, ke;    // ke: channels staged per pixel (multiple of 8)
  int OH, OW, Cso, C;    // C: logical big-side channels per class
  int GH, GW;            // anchor grid (ceil(OH/2), ceil(OW/2))
  int nhm, nwm, dh_min, dw_min, ntap;
  int KP, PP, wpitch;    // K per tap padded to 32; LDS pixel pitch; filter column pitch (elements)
  int TA, ntr;           // anchor rows per workgroup, row tiles per image
  int TW, ntc;           // anchor columns per workgroup, column tiles per row tile (wide images: the halo tile would not fit)
  int y_off;             // LDS byte offset of the halo tile
  unsigned long long* stamps;   // diagnostic build (-DTDG_STAMPS) only: per-wave phase boundaries; null otherwise
  FastDiv fd_vpp, fd_hc;
  int act, mask_mode, accumulate;
  float leak;
};

__global__ void __launch_bounds__(512) bwd_fused_kernel(const FusedBwdArgs a) {
  constexpr int NTHR = 512, NWAVE = NTHR / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* sW = reinterpret_cast<bf16_t*>(smem);
  bf16_t* sY = reinterpret_cast<bf16_t*>(smem + a.y_off);     // past the zero-filled tail of the filter's last wave instruction
  const int tid = threadIdx.x;
  const int tile = blockIdx.x / a.ntc, ct = blockIdx.x - tile * a.ntc;
  const int img = tile / a.ntr, rt = tile - img * a.ntr;
  const int a0 = rt * a.TA, c0 = ct * a.TW;
  const int HR = a.TA + a.nhm - 1, HC = a.TW + a.nwm - 1;     // halo tile (pixels)
  unsigned long long fs0 = 0, fs1 = 0, fs2 = 0, fs3 = 0, fs4 = 0;
  TDG_STAMP(fs0);

  // ---- stage the merged filter (straight copy) and the halo tile by LDS-DMA: every 16-byte chunk of both images
  // is one lane of a wave instruction (destination lane-linear), halo pixels outside the image, the pitch padding
  // and the slack behind the last pixel come from out-of-range sources (zero fill), so all loads are in flight at
  // once and one wait ends the staging (with one workgroup per CU it was latency-bound as load/store batches)
  const int wave = tid >> 6, lane = tid & 63;
  {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(a.w, (unsigned)(16 * a.wpitch * 2));
    const __amdgpu_buffer_rsrc_t rY = make_rsrc(a.y + (size_t)img * a.SH * a.SW * a.Cs, (unsigned)(a.SH * a.SW * a.Cs * 2));
    const int nvW = 16 * a.wpitch / 8;
    for (int g0 = wave * 64; g0 < nvW; g0 += NWAVE * 64) {
      const int g = g0 + lane;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)(smem + g0 * 16), 16, g < nvW ? (unsigned)g * 16u : OOB_OFFSET, 0, 0, 0);
    }
    const int vpp = a.PP / 8, vreal = a.ke / 8, npx = HR * HC;
    const int nvY = npx * vpp + a.KP / 8;
    char* sYb = reinterpret_cast<char*>(sY);
    for (int g0 = wave * 64; g0 < nvY; g0 += NWAVE * 64) {
      const int g = g0 + lane;
      const int pix = (int)fd_div((unsigned)g, a.fd_vpp), v = g - pix * vpp;
      const int hr = (int)fd_div((unsigned)pix, a.fd_hc), hc = pix - hr * HC;
      const int r = a0 + hr + a.dh_min, c = c0 + hc + a.dw_min;
      const bool ok = pix < npx && v < vreal && (unsigned)r < (unsigned)a.SH && (unsigned)c < (unsigned)a.SW;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rY, (lds_ptr_t)(sYb + g0 * 16), 16, ok ? (unsigned)(((r * a.SW + c) * a.Cs + v * 8) * 2) : OOB_OFFSET, 0, 0, 0);
    }
  }
  TDG_STAMP(fs1);
  __syncthreads();
  TDG_STAMP(fs2);

  const int r16 = lane & 15, q = lane >> 4;
  const int npix = a.TA * a.TW, ntile = (npix + 15) / 16;
  const int kcn = a.KP / 32;
  const bf16_t* wrow = sW + r16 * a.wpitch + q * 8;
  for (int t0 = wave; t0 < ntile; t0 += 2 * NWAVE) {
    // two 16-pixel tiles share every filter fragment
    const int t1 = t0 + NWAVE;
    const bool has1 = t1 < ntile;
    int p0 = t0 * 16 + r16, p1 = t1 * 16 + r16;
    const bool ok0 = p0 < npix, ok1 = has1 && p1 < npix;
    p0 = ok0 ? p0 : 0;
    p1 = ok1 ? p1 : 0;
    const int al0 = p0 / a.TW, b0 = p0 - al0 * a.TW;
    const int al1 = p1 / a.TW, b1 = p1 - al1 * a.TW;
    const bf16_t* base0 = sY + (size_t)(al0 * HC + b0) * a.PP + q * 8;
    const bf16_t* base1 = sY + (size_t)(al1 * HC + b1) * a.PP + q * 8;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int th = 0; th < a.nhm; ++th)
      for (int tw = 0; tw < a.nwm; ++tw) {
        const int toff = (th * HC + tw) * a.PP;
        const bf16_t* wt = wrow + (th * a.nwm + tw) * a.KP;
#pragma unroll 4
        for (int kc = 0; kc < kcn; ++kc) {
          const bf16x8 fw = *reinterpret_cast<const bf16x8*>(wt + kc * 32);
          const bf16x8 f0 = *reinterpret_cast<const bf16x8*>(base0 + toff + kc * 32);
          const bf16x8 f1 = *reinterpret_cast<const bf16x8*>(base1 + toff + kc * 32);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, f0, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, f1, acc1, 0, 0, 0);
        }
      }
#ifdef TDG_STAMPS
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    TDG_STAMP(fs3);
#endif
    // ---- epilogue: lane holds columns 4q..4q+3 = (class, c) of its pixel ------------------------------------
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool ok = h ? ok1 : ok0;
      if (!ok) continue;
      const f32x4 acc = h ? acc1 : acc0;
      const int ar = a0 + (h ? al1 : al0), bc = c0 + (h ? b1 : b0);
      if (ar >= a.GH || bc >= a.GW) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = q * 4 + e;
        const int cls = col / a.C, c = col - cls * a.C;
        if (cls >= 4) continue;
        const int oy = 2 * ar + (cls >> 1), ox = 2 * bc + (cls & 1);
        if (oy >= a.OH || ox >= a.OW) continue;
        const size_t o = (((size_t)img * a.OH + oy) * a.OW + ox) * a.Cso + c;
        float v = acc[e] + (a.bias ? a.bias[c] : 0.f);
        v = apply_act(v, a.act, a.leak);
        if (a.accumulate) v += (float)a.x[o];
        if (a.mask_mode != TDG_MASK_NONE) v *= mask_factor((float)a.mask_src[o], a.mask_mode, a.leak);
        a.x[o] = (bf16_t)v;
      }
    }
  }
#ifdef TDG_STAMPS
  if (a.stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TDG_STAMP(fs4);
    if (lane == 0) {
      unsigned long long* o = a.stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
      o[0] = fs0; o[1] = fs1; o[2] = fs2; o[3] = fs3; o[4] = fs4;
    }
  }
#endif
}

// ============================================================================================
// Stride-2 backward-data / conv2d_transpose onto a thin big side whose (tap, channel) pairs fit 64 GEMM columns
// (pix2pix: the decoder's last layer 128 -> 1 at 256 x 256, the critic's first layer gradient 64 -> 4): GEMM + col2im.
//   P[pixel][(tap, c)] = sum_k y[pixel][k] * w[tap][c][k]      one 16-column MFMA tile per 16 (tap, c) pairs: no zero-stuffed
//                                                              taps, no class padding (the fused-class kernel above spends
//                                                              9 union taps x 16 columns on 4 taps x 4 columns here)
//   x[2a + ph][2b + pw][c] = sum_{taps t of class (ph, pw)} P[(a + dh_t, b + dw_t)][(t, c)]
// A workgroup owns TA x TW anchors of one image: its waves multiply the (TA + halo) x (TW + halo) small-side pixels -- A
// fragments straight from global memory (each lane 16 bytes of one pixel; pixels outside the image are out-of-range
// offsets = zeros), the filter [columns][KP] from LDS -- park P as f32 in LDS and then sum each output pixel's taps.
// 40-50 KB of LDS: three workgroups per CU overlap one another's load and sum phases.   bf16 only.
// ============================================================================================
#define C2I_MAX_CLS_TAPS 9
struct Col2imArgs {
  const bf16_t* y;       // small side [n][SH][SW][Cs]
  const bf16_t* w;       // packed filter [NT * 16][KP]: row = tap * C + c
  bf16_t* x;             // big side [n][OH][OW][Cso]
  const float* bias;
  const bf16_t* mask_src;
  int SH, SW, Cs;
  int OH, OW, Cso, C;
  int KP, NT, pitch, ke; // k padded to 32; 16-column tiles; f32 words per pixel of P in LDS; channels read per pixel
  int TA, TW, ntr, ntc, HR, HC, dh_min, dw_min;
  int p_off;             // LDS byte offset of P
  int act, mask_mode, accumulate;
  float leak;
  int debug;                                // TDG_DEBUG_ABLATE (diagnostics): 1 = no global loads, 2 = no stores, 3 = no MFMA phase
  int cls_ntaps[4];                         // class = 2 * (oy & 1) + (ox & 1)
  int cls_tap[4][C2I_MAX_CLS_TAPS];         // (halo row offset << 16) | (halo column offset << 8) | filter tap
  FastDiv fd_hc, fd_c, fd_ow2;
};

template <int NT>
__global__ void __launch_bounds__(256) bwd_col2im_kernel(const Col2imArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* sW = reinterpret_cast<bf16_t*>(smem);
  float* sP = reinterpret_cast<float*>(smem + a.p_off);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tile = blockIdx.x / a.ntc, ct = blockIdx.x - tile * a.ntc;
  const int img = tile / a.ntr, rt = tile - img * a.ntr;
  const int a0 = rt * a.TA, c0 = ct * a.TW;
  const int npx = a.HR * a.HC, ntile = (npx + 15) / 16;
  // the class tap table (a per-lane index into kernel arguments would be a global load per tap) and the filter: straight copies
  int* sTab = reinterpret_cast<int*>(smem + a.p_off - 256);
  if (tid < 4) sTab[tid] = a.cls_ntaps[tid];
  if (tid < 4 * C2I_MAX_CLS_TAPS) sTab[4 + tid] = a.cls_tap[tid / C2I_MAX_CLS_TAPS][tid % C2I_MAX_CLS_TAPS];
  {
    const int nv = NT * 16 * a.KP / 8;
    const uint4* src = reinterpret_cast<const uint4*>(a.w);
    uint4* dst = reinterpret_cast<uint4*>(sW);
    for (int i = tid; i < nv; i += 256) dst[i] = src[i];
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rY = make_rsrc(a.y + (size_t)img * a.SH * a.SW * a.Cs, (unsigned)(a.SH * a.SW * a.Cs * 2));
  const int r16 = lane & 15, q = lane >> 4;
  const int kcn = a.KP / 32;
  const bf16_t* wrow = sW + r16 * a.KP + q * 8;
  for (int t0 = wave; t0 < (a.debug == 3 ? 0 : ntile); t0 += 8) {
    // two 16-pixel tiles share every filter fragment
    unsigned off[2];
    int hp[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int t = t0 + 4 * h;
      hp[h] = t * 16 + r16;
      const int hr = (int)fd_div((unsigned)hp[h], a.fd_hc), hc = hp[h] - hr * a.HC;
      const int r = a0 + hr + a.dh_min, c = c0 + hc + a.dw_min;
      const bool ok = t < ntile && hp[h] < npx && (unsigned)r < (unsigned)a.SH && (unsigned)c < (unsigned)a.SW;
      off[h] = (ok && a.debug != 1) ? (unsigned)(((r * a.SW + c) * a.Cs + q * 8) * 2) : OOB_OFFSET;
    }
    f32x4 acc[2][NT];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[h][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int kc = 0; kc < kcn; ++kc) {
      const bool kok = kc * 32 + q * 8 < a.ke;                 // (KP > ke: the next pixel's channels must not meet the zero filter rows as NaN)
      const bf16x8 f0 = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rY, (off[0] == OOB_OFFSET || !kok) ? OOB_OFFSET : off[0] + kc * 64, 0, 0));
      const bf16x8 f1 = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rY, (off[1] == OOB_OFFSET || !kok) ? OOB_OFFSET : off[1] + kc * 64, 0, 0));
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const bf16x8 fw = *reinterpret_cast<const bf16x8*>(wrow + n * 16 * a.KP + kc * 32);
        acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, f0, acc[0][n], 0, 0, 0);
        acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, f1, acc[1][n], 0, 0, 0);
      }
    }
    // lane holds columns 16n + 4q .. + 3 of its pixel
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (t0 + 4 * h < ntile && hp[h] < npx) {
#pragma unroll
        for (int n = 0; n < NT; ++n) *reinterpret_cast<f32x4*>(sP + (size_t)hp[h] * a.pitch + n * 16 + q * 4) = acc[h][n];
      }
  }
  __syncthreads();
  // ---- col2im: each output value of the tile is the sum of its class's taps ---------------------------------------
  const int ow2 = 2 * a.TW, nout = 4 * a.TA * a.TW * a.C;
  for (int o = tid; o < nout; o += 256) {
    const int pix = (int)fd_div((unsigned)o, a.fd_c), c = o - pix * a.C;
    const int oyl = (int)fd_div((unsigned)pix, a.fd_ow2), oxl = pix - oyl * ow2;
    const int oy = 2 * a0 + oyl, ox = 2 * c0 + oxl;
    if (oy >= a.OH || ox >= a.OW) continue;
    const int cls = 2 * (oyl & 1) + (oxl & 1), al = oyl >> 1, bl = oxl >> 1;
    float v = 0.f;
    const int nt = sTab[cls];
    for (int t = 0; t < nt; ++t) {
      const int pk = sTab[4 + cls * C2I_MAX_CLS_TAPS + t];
      v += sP[(size_t)((al + (pk >> 16)) * a.HC + bl + ((pk >> 8) & 0xff)) * a.pitch + (pk & 0xff) * a.C + c];
    }
    const size_t g = (((size_t)img * a.OH + oy) * a.OW + ox) * a.Cso + c;
    v = apply_act(v + (a.bias ? a.bias[c] : 0.f), a.act, a.leak);
    if (a.accumulate) v += (float)a.x[g];
    if (a.mask_mode != TDG_MASK_NONE) v *= mask_factor((float)a.mask_src[g], a.mask_mode, a.leak);
    if (a.debug != 2 || v == 12345.f) a.x[g] = (bf16_t)v;
  }
}

// ============================================================================================
// Forward conv to ONE output channel (the PatchGAN critic's logits: 512 -> 1, 4x4, M = images x 8 x 8): a dot product
// per output pixel.  The 128 x 16 MFMA tile spends a 128-step K loop per workgroup on 64 workgroups (0.11 ms for 34 MB
// of input); here one wave owns one output pixel, lanes take 16-byte channel chunks of every tap (padding taps are
// out-of-range offsets), the packed filter row is read the same way.
// ============================================================================================
struct ConvN1Args {
  const void* x;
  const void* w;         // packed forward filter: one row [ntaps][C] in the launch's tap order
  void* y;
  const float* bias;
  unsigned x_bytes, w_bytes;
  int SH, SW, Cs, C, OH, OW, Cso, stride, ntaps, M;
  int act;
  float leak;
  int tap[IG_MAX_TAPS];
  FastDiv fd_ohw, fd_ow;
};

template <typename T>
__global__ void __launch_bounds__(256) conv_n1_fwd_kernel(const ConvN1Args a) {
  constexpr int VEC = 16 / (int)sizeof(T);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + wave;
  if (m >= a.M) return;
  const int n = (int)fd_div((unsigned)m, a.fd_ohw), rem = m - n * a.OH * a.OW;
  const int oy = (int)fd_div((unsigned)rem, a.fd_ow), ox = rem - oy * a.OW;
  const __amdgpu_buffer_rsrc_t rX = make_rsrc(a.x, a.x_bytes);
  const __amdgpu_buffer_rsrc_t rW = make_rsrc(a.w, a.w_bytes);
  const int cv = a.C / VEC;
  float s = 0.f;
#pragma unroll 4
  for (int t = 0; t < a.ntaps; ++t) {
    const int pk = a.tap[t];
    const int ih = oy * a.stride + tap_dh(pk), iw = ox * a.stride + tap_dw(pk);
    const bool ok = (unsigned)ih < (unsigned)a.SH && (unsigned)iw < (unsigned)a.SW;
    const unsigned xb = (unsigned)(((n * a.SH + ih) * a.SW + iw) * a.Cs) * (unsigned)sizeof(T);
    const unsigned wb = (unsigned)(t * a.C) * (unsigned)sizeof(T);
    for (int ch = lane; ch < cv; ch += 64) {
      const auto xv = __builtin_amdgcn_raw_buffer_load_b128(rX, ok ? xb + ch * 16u : OOB_OFFSET, 0, 0);
      const auto wv = __builtin_amdgcn_raw_buffer_load_b128(rW, wb + ch * 16u, 0, 0);
      if constexpr (sizeof(T) == 2) {
        const bf16x8 xf = __builtin_bit_cast(bf16x8, xv), wf = __builtin_bit_cast(bf16x8, wv);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)xf[e] * (float)wf[e];
      } else {
        const f32x4 xf = __builtin_bit_cast(f32x4, xv), wf = __builtin_bit_cast(f32x4, wv);
#pragma unroll
        for (int e = 0; e < 4; ++e) s += xf[e] * wf[e];
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0)
    static_cast<T*>(a.y)[(size_t)m * a.Cso] = from_f32<T>(apply_act(s + (a.bias ? a.bias[0] : 0.f), a.act, a.leak));
}

// ============================================================================================
// Forward conv of a THIN input (C <= 4 channels: the discriminator's / encoder's first layer on images).  The
// implicit-GEMM kernels see such an input as 8-channel pixels, i.e. K = taps x 8 padded to 64s (75 real of 256 for
// 5x5x3: that GEMM is bound by its padded MFMA work).  Here K is compact, k = tap * C + c padded to 32 (75 -> 96):
// a workgroup stages the input patch of its 128 output pixels (one or more whole output rows of one image, zeros
// outside the image) and the filter tile [224][K] in LDS; A fragments are gathered from the patch element by
// element through a k -> patch-offset table, B fragments are 16-byte rows.  4 waves as 2 x 2, 64 pixels x 112
// columns per wave (7 column tiles; the 14th of the workgroup is spill), LDS-staged 16-byte epilogue.   bf16 only.
// ============================================================================================
#define TH_PIX 128
struct ThinFwdArgs {
  const bf16_t* x;        // [n][H][W][Cs]
  const bf16_t* w;        // packed [ntiles_n * 224][WP]
  bf16_t* y;              // [n][OH][OW][Cso]
  const float* bias;
  const bf16_t* mask_src;
  int H, W, Cs, C;
  int OH, OW, Cso, N;
  int KH, KW, stride, pad_t, pad_l;
  int Kp, WP;             // K padded to 32; filter row pitch (elements, odd number of 16-byte chunks)
  int TH, tiles_per_image, ntiles_n;
  int PH, PW;             // staged patch (pixels), 4 elements per pixel
  int y_off, k_off;       // LDS byte offsets of the patch and of the k table
  int act, mask_mode;
  float leak;
  int debug;              // TDG_DEBUG_ABLATE (diagnostics): 7 = no global stores, 8 = no MFMA loop
  unsigned long long* stamps;   // diagnostic build (-DTDG_STAMPS) only: per-wave phase boundaries; null otherwise
  FastDiv fd_ow, fd_c, fd_pw;
};

// NC = 208: 2 x 2 waves of 64 pixels x 112 columns (the GAN critic's c1: N = 200);  NC = 64: 4 x 1 waves of 32 pixels x 64
// columns (pix2pix e1 / m1, N = 64: the 128 x 64 implicit-GEMM tile spent its life in ring prologue and epilogue latency:
// 1.5 TB/s of output writes)
template <int NC>
__global__ void __launch_bounds__(256, 2) thin_fwd_kernel(const ThinFwdArgs a) {
  constexpr int BNL = NC == 208 ? 224 : NC, TM = NC == 208 ? 4 : 2, TN = NC == 208 ? 7 : NC / 16;
  constexpr int WROWS = TM * 16;                               // pixels per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* sW = reinterpret_cast<bf16_t*>(smem);
  unsigned short* sP = reinterpret_cast<unsigned short*>(smem + a.y_off);          // patch [PH][PW][4]
  unsigned short* sK = reinterpret_cast<unsigned short*>(smem + a.k_off);           // k -> patch element offset [Kp]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r16 = lane & 15, q = lane >> 4;
  const int tile_n = blockIdx.x % a.ntiles_n, tile_m = blockIdx.x / a.ntiles_n;
  const int img = tile_m / a.tiles_per_image, rt = tile_m - img * a.tiles_per_image;
  const int oy0 = rt * a.TH, n0 = tile_n * NC;
  const int npix = min(a.TH, a.OH - oy0) * a.OW;
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0;
  TDG_STAMP(ts0);

  // ---- stage by LDS-DMA (every load in flight at once; as load / store batches the staging was latency-bound):
  // filter tile = straight copy in 16-byte chunks; patch = 4-byte pieces (two channels), out-of-image pieces and
  // the zero element come from out-of-range sources; then the k -> offset table
  {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(a.w + (size_t)tile_n * BNL * a.WP, (unsigned)(BNL * a.WP * 2));
    const __amdgpu_buffer_rsrc_t rX = make_rsrc(a.x + (size_t)img * a.H * a.W * a.Cs, (unsigned)(a.H * a.W * a.Cs * 2));
    const int nvW = BNL * a.WP / 8;
    for (int g0 = wave * 64; g0 < nvW; g0 += 4 * 64) {
      const int g = g0 + lane;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)(smem + g0 * 16), 16, g < nvW ? (unsigned)g * 16u : OOB_OFFSET, 0, 0, 0);
    }
    const int iy0 = oy0 * a.stride - a.pad_t, ix0 = -a.pad_l;
    const int nP = a.PH * a.PW * 2 + 4;                           // 4-byte pieces incl. the zero element's
    char* sPb = reinterpret_cast<char*>(sP);
    for (int g0 = wave * 64; g0 < nP; g0 += 4 * 64) {
      const int g = g0 + lane;
      const int p = g >> 1, half = g & 1;
      const int py = (int)fd_div((unsigned)p, a.fd_pw), px = p - py * a.PW;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = p < a.PH * a.PW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_ptr_t)(sPb + g0 * 4), 4, ok ? (unsigned)(((iy * a.W + ix) * a.Cs + half * 2) * 2) : OOB_OFFSET, 0, 0, 0);
    }
    for (int k = tid; k < a.Kp; k += 256) {
      const int tap = (int)fd_div((unsigned)k, a.fd_c), c = k - tap * a.C;
      const int kh = tap / a.KW, kw = tap - kh * a.KW;
      sK[k] = tap < a.KH * a.KW ? (unsigned short)((kh * a.PW + kw) * 4 + c) : (unsigned short)0xffff;
    }
  }
  __syncthreads();
  TDG_STAMP(ts1);

  const int wm = NC == 208 ? wave >> 1 : wave, wn = NC == 208 ? wave & 1 : 0;
  // the wave's bias pieces, loaded now so that their latency hides behind the MFMA loop (N % 4 == 0: plan_fwd_thin)
  f32x4 bvs[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) bvs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (a.bias) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 16 + q * 4;
      const f32x4 t = *reinterpret_cast<const f32x4*>(a.bias + min(n, a.N - 4));
      if (n < a.N) bvs[j] = t;
    }
  }
  // patch element offset of each of this lane's 4 pixels (row r16 of row-fragment i)
  int pbase[TM];
  const int zero_off = a.PH * a.PW * 4;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    int p = wm * WROWS + i * 16 + r16;
    p = p < npix ? p : 0;
    const int oy = (int)fd_div((unsigned)p, a.fd_ow), ox = p - oy * a.OW;
    pbase[i] = (oy * a.stride * a.PW + ox * a.stride) * 4;
  }
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16_t* wrow = sW + (size_t)(wn * TN * 16 + r16) * a.WP + q * 8;

  for (int ks = 0; ks < (a.debug == 8 ? 0 : a.Kp / 32); ++ks) {
    // this lane's 8 k offsets of the step (the same for every pixel)
    const i32x4 ko = *reinterpret_cast<const i32x4*>(sK + ks * 32 + q * 8);
    int off[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned lo = (unsigned)ko[e] & 0xffffu, hi = (unsigned)ko[e] >> 16;
      off[2 * e] = lo == 0xffffu ? -1 : (int)lo;
      off[2 * e + 1] = hi == 0xffffu ? -1 : (int)hi;
    }
    bf16x8 fa[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      unsigned pk[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned lo = sP[off[2 * e] < 0 ? zero_off : pbase[i] + off[2 * e]];
        const unsigned hi = sP[off[2 * e + 1] < 0 ? zero_off : pbase[i] + off[2 * e + 1]];
        pk[e] = lo | (hi << 16);
      }
      const i32x4 t = {(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]};
      fa[i] = __builtin_bit_cast(bf16x8, t);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const bf16x8 fb = *reinterpret_cast<const bf16x8*>(wrow + (size_t)j * 16 * a.WP + ks * 32);
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa[i], acc[i][j], 0, 0, 0);
    }
  }
  TDG_STAMP(ts2);
  __syncthreads();                                        // every wave is done with sW / sP: reuse LDS as the staging tile

  // ---- epilogue: bias + activation into an LDS tile, then whole 16-byte chunks of pixel rows to HBM ----------------
  constexpr int PE = NC * 2 + 16, CPR = NC * 2 / 16;
  char* sE = smem;
  dispatch_act(a.act, [&](auto tag) {
    constexpr int ACT = decltype(tag)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = (wn * TN + j) * 16 + q * 4;
      if (col >= NC) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        f32x4 v = acc[i][j] + bvs[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act_c<ACT>(v[e], a.act, a.leak);
        *reinterpret_cast<bf16x4*>(sE + (wm * WROWS + i * 16 + r16) * PE + col * 2) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      }
    }
  });
  __syncthreads();
  TDG_STAMP(ts3);
  bf16_t* yi = a.y + ((size_t)img * a.OH + oy0) * a.OW * a.Cso;          // the tile's pixels are contiguous rows of y
  const float mlow = mask_low(a.mask_mode, a.leak);
  for (int c = tid; c < npix * CPR; c += 256) {
    const int row = c / CPR, cc = c - row * CPR;
    const int n = n0 + cc * 8;
    if (n >= a.N || a.debug == 7) continue;
    bf16x8 v = *reinterpret_cast<const bf16x8*>(sE + row * PE + cc * 16);
    const size_t o = (size_t)row * a.Cso + n;
    if (a.mask_mode != TDG_MASK_NONE) {
      const bf16x8 mv = *reinterpret_cast<const bf16x8*>(a.mask_src + ((size_t)img * a.OH + oy0) * a.OW * a.Cso + o);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] * ((float)mv[e] > 0.f ? 1.f : mlow));
    }
    *reinterpret_cast<bf16x8*>(yi + o) = v;
  }
#ifdef TDG_STAMPS
  if (a.stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TDG_STAMP(ts4);
    if (lane == 0) {
      unsigned long long* o = a.stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
      o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3; o[4] = ts4;
      unsigned hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      o[5] = hw;
    }
  }
#endif
}

// ============================================================================================
// host side: planning + launch
// ============================================================================================
namespace {

thread_local double t_flops = 0.0;   // algorithmic FLOPs of the entry-point call being dispatched (for tdg_timing_*)
thread_local const TdgEpilogue* t_col = nullptr;   // column-partial request of the call being dispatched (fill_epilogue)
thread_local const TdgEpilogue* t_splitk = nullptr;   // epilogue carrying a split-K workspace (fill_epilogue)

struct TileCfg { int bm, bn; };

inline int pick_bn(int n) {
  if (n <= 16) return 16;
  if (n <= 64) return 64;
  const double w128 = (double)tdg_round_up(n, 128) / n, w208 = (double)tdg_round_up(n, 208) / n;
  return w208 < w128 - 1e-9 ? 208 : 128;
}

template <typename T, int BM, int BN, int WGM, int WGN>
int launch_fwd_cfg(const IgArgs& a, bool veca, int grid_x, int nclasses, hipStream_t s) {
  const size_t lds = (size_t)(BM + BN) * IG_BKB + IG_MAX_TAPS * sizeof(int);
  dim3 grid(grid_x, 1, nclasses), block(256);
  static char name[64] = "";
  if (!name[0]) snprintf(name, sizeof(name), "igemm_fwd_kernel<%s,%d,%d>", sizeof(T) == 2 ? "bf16" : "f32", BM, BN);
  tdg_note_kernel(name);
  tdg_timing_start(name, t_flops, s);
  if (veca)
    hipLaunchKernelGGL((igemm_fwd_kernel<T, BM, BN, WGM, WGN, true>), grid, block, lds, s, a);
  else
    hipLaunchKernelGGL((igemm_fwd_kernel<T, BM, BN, WGM, WGN, false>), grid, block, lds, s, a);
  tdg_timing_stop(s);
  TDG_HIP_LAUNCH_CHECK("igemm_fwd");
  return TDG_OK;
}

// columns [n_begin, n_begin + ntiles_n * BN) of the problem (clipped to N)
template <typename T, int BM, int BN, int NS, int NW = 8, int WS = 0>
int launch_fwd_dma(IgArgs& a, int mmax, hipStream_t s, int n_begin = 0, int ntiles_n = -1) {
  a.n_begin = n_begin;
  a.ntiles_n = ntiles_n < 0 ? tdg_ceil_div(a.N, BN) : ntiles_n;
  a.ntiles_m_max = tdg_ceil_div(mmax, BM);
  if (t_col && sizeof(T) == 2 && !a.accumulate && (a.N & 3) == 0 && n_begin == 0 && a.ntiles_n * BN >= a.N) {
    const int nblk = a.nclasses * a.ntiles_m_max;
    if ((size_t)nblk * 2 * a.N * sizeof(float) <= t_col->col_partial_bytes) {
      a.col_partial = t_col->col_partial;
      a.col_mode = t_col->col_mode;
      a.col_images = t_col->col_images;
      *t_col->col_nblk_out = nblk;
    }
  }
  constexpr int BNL = 2 * ((BN / 16 + 1) / 2) * 16;
  const size_t lds = NS * (size_t)(BM + BNL) * IG_BKB + IG_MAX_TAPS * sizeof(int) + BNL * sizeof(float);   // ring, taps, bias row
  static_assert(NS * (size_t)(BM + BNL) * IG_BKB + IG_MAX_TAPS * sizeof(int) + BNL * sizeof(float) <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_fwd_dma_kernel<T, BM, BN, NS, NW, WS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  static char name[64] = "";
  if (!name[0]) snprintf(name, sizeof(name), WS ? "igemm_fwd_dma_kernel<%s,%d,%d,%d,8,1>" : (NW == 8 ? "igemm_fwd_dma_kernel<%s,%d,%d,%d>" : "igemm_fwd_dma_kernel<%s,%d,%d,%d,4,0>"), sizeof(T) == 2 ? "bf16" : "f32", BM, BN, NS);
  tdg_note_kernel(name);
  const int n_end = n_begin + a.ntiles_n * BN < a.N ? n_begin + a.ntiles_n * BN : a.N;
  // split-K: a grid that leaves most of the chip idle (pix2pix's 1x1 ... 8x8 bottleneck layers: 4 - 32 workgroups, each
  // walking 128 - 256 K steps) is cut along K into f32 partial tiles, finished by splitk_finish_kernel
  a.ksplit = 1;
  a.steps_per_split = 1 << 30;
  a.slab = nullptr;
  a.slab_rows = a.ntiles_m_max * BM;
  int smax = 0;
  for (int c = 0; c < a.nclasses; ++c) smax = a.cls[c].nsteps > smax ? a.cls[c].nsteps : smax;
  const long long wgs = (long long)a.ntiles_n * a.ntiles_m_max * a.nclasses;
  const int ks_force = getenv("TDG_KSPLIT") ? atoi(getenv("TDG_KSPLIT")) : 0;                  // diagnostics: 1 = never, n = n splits
  if (!WS && t_splitk && t_splitk->splitk_ws && wgs <= 128 && smax >= 16 && n_begin == 0 && a.ntiles_n * BN >= a.N && ks_force != 1) {
    int want = (int)(256 / wgs);
    if (want > smax / 4) want = smax / 4;                    // >= 4 steps per split
    if (want > 8) want = 8;                                  // measured (64 images, 512 -> 512, 4x4 s2): 2x2 / 4x4 inputs 0.019 / 0.020 ms at 8 splits, 0.026 / 0.032 at 32
    if (ks_force > 1) want = ks_force;
    const int per = tdg_ceil_div(smax, want);
    const int nsplit = tdg_ceil_div(smax, per);
    const size_t need = (size_t)nsplit * a.nclasses * a.slab_rows * a.N * sizeof(float);
    if (nsplit > 1 && need <= t_splitk->splitk_ws_bytes) {
      a.ksplit = nsplit;
      a.steps_per_split = per;
      a.slab = static_cast<float*>(t_splitk->splitk_ws);
      a.col_partial = nullptr;                               // (the partial tiles are not the stored tile)
      if (t_col) *t_col->col_nblk_out = 0;
    }
  }
  dim3 grid(a.ntiles_n * a.ntiles_m_max, a.ksplit, a.nclasses), block(64 * NW);
  tdg_timing_start(name, t_flops * (double)(n_end - n_begin) / (double)a.N, s);
  hipLaunchKernelGGL((igemm_fwd_dma_kernel<T, BM, BN, NS, NW, WS>), grid, block, lds, s, a);
  if (a.ksplit > 1) {
    int mmx = 0;
    for (int c = 0; c < a.nclasses; ++c) mmx = a.cls[c].M > mmx ? a.cls[c].M : mmx;
    const long long items = (long long)mmx * ((a.N + 3) / 4);
    hipLaunchKernelGGL((splitk_finish_kernel<T>), dim3((unsigned)((items + 255) / 256), 1, a.nclasses), dim3(256), 0, s, a);
  }
  tdg_timing_stop(s);
  TDG_HIP_LAUNCH_CHECK("igemm_fwd_dma");
  return TDG_OK;
}

// igemm_fwd_patch_kernel: does it apply to this launch?  Fills the tap groups / lattice of every class when it does.
// Conditions (each one is what the kernel's indexing assumes): bf16 vector gather, CK-chunk K slices, BN-column tiles with
// the staged epilogue, row tiles of whole images OR of one bh x bw block of an image's anchor grid (then the classes' GH, GW,
// fd_ghw, fd_gw are rewritten to the block's and nbh, nbw, halo set -- only when the whole plan holds), a source whose
// sub-lattices all have the same shape, patches that fit PIECES KiB, tap groups in K order, and a phase schedule in which
// every patch lands two steps before its first read (the loader's own rule, simulated here).
template <int BM, int BN = 208, int CK = PT_CK, int PIECES = PT_PIECES, bool FORCE_HALO = false>
bool plan_fwd_patch(IgArgs& a, int mmax) {
  const int enabled = getenv("TDG_PATCH") ? atoi(getenv("TDG_PATCH")) : 1;     // diagnostics: 0 = igemm_fwd_dma_kernel, 2 = also on small grids
  if (!enabled || (int)a.fd_ck.d != CK || a.accumulate || (a.N & 3) || (a.Cso & 3)) return false;
  if (a.sigma < 1 || a.sigma > 2 || a.SH % a.sigma || a.SW % a.sigma) return false;
  const int LH = a.SH / a.sigma, LWd = a.SW / a.sigma;        // the source's sub-lattice
  if (enabled != 2 && (long long)tdg_ceil_div(mmax, BM) * tdg_ceil_div(a.N, BN) * a.nclasses < 160) return false;   // small grids: the 112-column forms
  struct Geo { int bh, bw, nbh, nbw, halo, QH, QW; } geo[IG_MAX_CLASSES];
  for (int ci = 0; ci < a.nclasses; ++ci) {
    IgClass& c = a.cls[ci];
    if (c.nbh * c.nbw > 1) return false;                       // (already rewritten: a plan is made once per launch)
    const int ghw = c.GH * c.GW;
    if (ghw <= 0 || c.M % ghw) return false;
    if (c.nsteps < 4 || c.nsteps * 64 > 16384 || c.ntaps > 31) return false;
    // tap groups: consecutive taps with equal (dh mod sigma, dw mod sigma); the lattice offsets of the taps
    int ng = 0, dmax = 0;
    for (int t = 0; t < c.ntaps; ++t) {
      const int dh = (signed char)(c.tap[t] & 0xff), dw = (signed char)((c.tap[t] >> 8) & 0xff);
      const int ph = ((dh % a.sigma) + a.sigma) % a.sigma, pw = ((dw % a.sigma) + a.sigma) % a.sigma;
      const int dhq = (dh - ph) / a.sigma, dwq = (dw - pw) / a.sigma;
      if (dhq < -8 || dhq > 7 || dwq < -8 || dwq > 7) return false;
      dmax = abs(dhq) > dmax ? abs(dhq) : dmax;
      dmax = abs(dwq) > dmax ? abs(dwq) : dmax;
      if (ng && c.grp[ng - 1].ph == ph && c.grp[ng - 1].pw == pw) { ++c.grp[ng - 1].nt; continue; }
      for (int g = 0; g < ng; ++g)
        if (c.grp[g].ph == ph && c.grp[g].pw == pw) return false;     // a group split in two: not the K order this kernel wants
      if (ng == 4) return false;
      c.grp[ng].t0 = t; c.grp[ng].nt = 1; c.grp[ng].ph = ph; c.grp[ng].pw = pw;
      ++ng;
    }
    c.ngroups = ng;
    Geo& g = geo[ci];
    if (BM % ghw == 0 && !(FORCE_HALO && ghw == BM)) {         // whole images: the patch is the image's sub-lattice, taps outside it are masked
      g = Geo{c.GH, c.GW, 1, 1, 0, LH, LWd};
    } else {                                                   // a block of one image with a halo of dmax lattice pixels
      int bw = c.GW < 16 ? c.GW : 16;
      while (bw > 1 && (c.GW % bw || BM % bw)) --bw;
      const int bh = BM / bw;
      if (bh * bw != BM || c.GH % bh || c.GW % bw) return false;
      g = Geo{bh, bw, c.GH / bh, c.GW / bw, dmax, bh + 2 * dmax, bw + 2 * dmax};
    }
    if (g.QW > 32 || g.QH > 64) return false;
    if ((BM / (g.bh * g.bw)) * g.QH * g.QW * PT_PIXB > PIECES * 1024 || g.bh > g.QH + 8 || g.bw > g.QW + 8) return false;
    for (int t = 0; t < c.ntaps; ++t) {
      const int dh = (signed char)(c.tap[t] & 0xff), dw = (signed char)((c.tap[t] >> 8) & 0xff);
      const int ph = ((dh % a.sigma) + a.sigma) % a.sigma, pw = ((dw % a.sigma) + a.sigma) % a.sigma;
      const int off = ((dh - ph) / a.sigma) * g.QW + (dw - pw) / a.sigma;
      if (off < -127 || off > 127) return false;
    }
    // the loader's schedule: phase p (>= PT_NPB) is issued in steps s_p, s_p + 1 and is visible from step s_p + 3
    const int P = a.nslices * ng, slc = c.ntaps * CK;
    auto first_chunk = [&](int p) { return (p / ng) * slc + c.grp[p % ng].t0 * CK; };
    auto end_chunk = [&](int p) { return (p / ng) * slc + (c.grp[p % ng].t0 + c.grp[p % ng].nt) * CK; };
    // patches 1 .. PT_NPB - 1 go out whole in steps 0 .. PT_NPB - 2 (visible to the reads of step issue + 2)
    for (int p = 1; p < PT_NPB && p < P; ++p)
      if (first_chunk(p) / 8 < (p - 1) + 2) return false;
    int prev_issue = PT_NPB - 3;                             // (the first two-step load starts at step PT_NPB - 1 at the earliest)
    for (int p = PT_NPB; p < P; ++p) {
      const int ready = (end_chunk(p - PT_NPB) - 1) / 8 + 1;
      const int issue = ready > prev_issue + 2 ? ready : prev_issue + 2;
      if (first_chunk(p) / 8 < issue + 3) return false;
      prev_issue = issue;
    }
  }
  for (int ci = 0; ci < a.nclasses; ++ci) {                    // the plan holds: commit the geometry
    IgClass& c = a.cls[ci];
    const Geo& g = geo[ci];
    c.QH = g.QH; c.QW = g.QW;
    c.fd_qhw = make_fastdiv(g.QH * g.QW); c.fd_qw = make_fastdiv(g.QW);
    c.nbh = g.nbh; c.nbw = g.nbw; c.halo = g.halo;
    if (g.nbh * g.nbw > 1) {
      c.GH = g.bh; c.GW = g.bw;
      c.fd_ghw = make_fastdiv(g.bh * g.bw); c.fd_gw = make_fastdiv(g.bw);
    }
  }
  return true;
}

template <int BM, int BN, int CK = PT_CK, int PIECES = PT_PIECES>
int launch_fwd_patch(IgArgs& a, int mmax, hipStream_t s) {
  a.n_begin = 0;
  a.ntiles_n = tdg_ceil_div(a.N, BN);
  a.ntiles_m_max = tdg_ceil_div(mmax, BM);
  if (t_col) {
    const int nblk = a.nclasses * a.ntiles_m_max;
    if ((size_t)nblk * 2 * a.N * sizeof(float) <= t_col->col_partial_bytes) {
      a.col_partial = t_col->col_partial;
      a.col_mode = t_col->col_mode;
      a.col_images = t_col->col_images;
      *t_col->col_nblk_out = nblk;
    }
  }
  a.ksplit = 1;
  a.steps_per_split = 1 << 30;
  a.slab = nullptr;
  int smax = 0;
  for (int c = 0; c < a.nclasses; ++c) smax = a.cls[c].nsteps > smax ? a.cls[c].nsteps : smax;
  constexpr int BNL = 2 * ((BN / 16 + 1) / 2) * 16;
  const size_t lds = PT_ZEROB + 3 * (size_t)BNL * IG_BKB + PT_NPB * (size_t)PIECES * 1024 + (size_t)smax * 64 + BNL * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_fwd_patch_kernel<BM, BN, 0, CK, PIECES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#ifdef TDG_STAMPS
    if constexpr (BM == 192) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_fwd_patch_kernel<BM, BN, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_fwd_patch_kernel<BM, BN, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_fwd_patch_kernel<BM, BN, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_fwd_patch_kernel<BM, BN, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
#endif
    attr_set = true;
  }
  static char name[64] = "";
  if (!name[0]) snprintf(name, sizeof(name), "igemm_fwd_patch_kernel<bf16,%d,%d>", BM, BN);
  tdg_note_kernel(name);
  dim3 grid(a.ntiles_n * a.ntiles_m_max, 1, a.nclasses), block(512);
  if (lds > 160 * 1024) {
    tdg_set_error("igemm_fwd_patch: %zu bytes of LDS", lds);
    return TDG_EUNSUPPORTED;
  }
  tdg_timing_start(name, t_flops, s);
#ifdef TDG_STAMPS
  // ablation instantiations (their results are garbage): compiled into the diagnostic library (build.sh stamps) only
  const int abl = (BM == 192 && getenv("TDG_PATCH_ABL")) ? atoi(getenv("TDG_PATCH_ABL")) : 0;
  if constexpr (BM == 192) {
    if (abl == 1) hipLaunchKernelGGL((igemm_fwd_patch_kernel<BM, BN, 1>), grid, block, lds, s, a);
    else if (abl == 2) hipLaunchKernelGGL((igemm_fwd_patch_kernel<BM, BN, 2>), grid, block, lds, s, a);
    else if (abl == 3) hipLaunchKernelGGL((igemm_fwd_patch_kernel<BM, BN, 3>), grid, block, lds, s, a);
    else if (abl == 4) hipLaunchKernelGGL((igemm_fwd_patch_kernel<BM, BN, 4>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((igemm_fwd_patch_kernel<BM, BN, 0>), grid, block, lds, s, a);
  } else {
    hipLaunchKernelGGL((igemm_fwd_patch_kernel<BM, BN, 0, CK, PIECES>), grid, block, lds, s, a);
  }
#else
  hipLaunchKernelGGL((igemm_fwd_patch_kernel<BM, BN, 0, CK, PIECES>), grid, block, lds, s, a);
#endif
  tdg_timing_stop(s);
  TDG_HIP_LAUNCH_CHECK("igemm_fwd_patch");
  return TDG_OK;
}

// igemm_fwd_bp_kernel: the block-patch plan (on a copy: the classes are rewritten only when everything holds) plus what the
// all-wave kernel assumes on top: one 16 x 16 block per row tile with a halo'd patch, tap groups of exactly four, a K of whole
// two-step phases.
template <int BN>
bool plan_fwd_bp(IgArgs& a, int mmax) {
  const char* e = getenv("TDG_BP");                         // variant tests: 0 = the 4 + 4 wave-specialised patch kernel
  if (e && atoi(e) == 0) return false;
  IgArgs t = a;
  if (!plan_fwd_patch<256, BN, 4, BP_PIECES, true>(t, mmax)) return false;
  for (int ci = 0; ci < t.nclasses; ++ci) {
    const IgClass& c = t.cls[ci];
    if (c.GH * c.GW != 256 || c.halo < 1 || c.QH * c.QW * PT_PIXB > BP_PIECES * 1024) return false;
    if (c.ntaps % 4 || c.ntaps != 4 * c.ngroups) return false;
    for (int g = 0; g < c.ngroups; ++g)
      if (c.grp[g].nt != 4) return false;
    if (c.nsteps != 2 * t.nslices * c.ngroups || c.M % 256) return false;
  }
  a = t;
  return true;
}

template <int BN>
int launch_fwd_bp(IgArgs& a, int mmax, hipStream_t s) {
  a.n_begin = 0;
  a.ntiles_n = tdg_ceil_div(a.N, BN);
  a.ntiles_m_max = tdg_ceil_div(mmax, 256);
  if (t_col) {
    const int nblk = a.nclasses * a.ntiles_m_max;
    if ((size_t)nblk * 2 * a.N * sizeof(float) <= t_col->col_partial_bytes) {
      a.col_partial = t_col->col_partial;
      a.col_mode = t_col->col_mode;
      a.col_images = t_col->col_images;
      *t_col->col_nblk_out = nblk;
    }
  }
  a.ksplit = 1;
  a.steps_per_split = 1 << 30;
  a.slab = nullptr;
  int smax = 0;
  for (int c = 0; c < a.nclasses; ++c) smax = a.cls[c].nsteps > smax ? a.cls[c].nsteps : smax;
  const size_t lds = PT_ZEROB + 3 * (size_t)BN * IG_BKB + PT_NPB * (size_t)BP_PIECES * 1024 + 1024 + (size_t)smax * 8 + BN * sizeof(float);
  if (lds > 160 * 1024) {
    tdg_set_error("igemm_fwd_bp: %zu bytes of LDS", lds);
    return TDG_EUNSUPPORTED;
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_fwd_bp_kernel<BN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
#ifdef TDG_STAMPS
  a.stamps = getenv("TDG_STAMP_PTR") ? (unsigned long long*)strtoull(getenv("TDG_STAMP_PTR"), nullptr, 0) : nullptr;
#endif
  static char name[64] = "";
  if (!name[0]) snprintf(name, sizeof(name), "igemm_fwd_bp_kernel<bf16,256,%d>", BN);
  tdg_note_kernel(name);
  dim3 grid(a.ntiles_n * a.ntiles_m_max, 1, a.nclasses), block(512);
  tdg_timing_start(name, t_flops, s);
  hipLaunchKernelGGL((igemm_fwd_bp_kernel<BN>), grid, block, lds, s, a);
  tdg_timing_stop(s);
  TDG_HIP_LAUNCH_CHECK("igemm_fwd_bp");
  return TDG_OK;
}

template <typename T>
int launch_fwd(IgArgs& a, bool veca, int bn, hipStream_t s) {
  constexpr int BM = 128;
#ifdef TDG_STAMPS
  static const int dbg = getenv("TDG_DEBUG_ABLATE") ? atoi(getenv("TDG_DEBUG_ABLATE")) : 0;   // loop ablations (garbage results): diagnostic library only
#else
  const int dbg = 0;
#endif
  const char* dma_env = getenv("TDG_DMA");          // diagnostics: 0 = never, 3 / 4 = force the 192 / 256-row tile
  const int dma_mode = dma_env ? atoi(dma_env) : 1;
  a.debug = dbg;
#ifdef TDG_STAMPS
  a.stamps = getenv("TDG_STAMP_PTR") ? (unsigned long long*)strtoull(getenv("TDG_STAMP_PTR"), nullptr, 0) : nullptr;
#endif
  int mmax = 0;
  for (int c = 0; c < a.nclasses; ++c) mmax = a.cls[c].M > mmax ? a.cls[c].M : mmax;
  // large problems: 256-row tiles fed by LDS-DMA (needs >= ~1 workgroup per CU to pay off)
  // 208-column problems with a vector gather always take the LDS-DMA kernel (measured faster than the
  // register-staged one even when the grid does not fill the chip)
  if constexpr (sizeof(T) == 2) {
    // whole-image row tiles with the gathered operand resident in LDS (igemm_fwd_patch_kernel), 192 or 128 rows: the
    // same cost model as below (rounds of 256 one-per-CU workgroups x rows per tile / relative efficiency; measured on the
    // 1536-image GEMMs: 8 - 13 % less time than igemm_fwd_dma_kernel's 192-row form)
    if (veca && bn == 208 && dma_mode == 1 && !getenv("TDG_DMA_BM") && !getenv("TDG_DMA_NW")) {
      const long long per = (long long)tdg_ceil_div(a.N, 208) * a.nclasses;
      const double c192 = (double)tdg_ceil_div(per * tdg_ceil_div(mmax, 192), 256) * 192 / 1.0;
      const double c128 = (double)tdg_ceil_div(per * tdg_ceil_div(mmax, 128), 256) * 128 / 0.75;   // in the step: 890 - 915 TF against 1170 - 1370
      const int pbm = getenv("TDG_PATCH_BM") ? atoi(getenv("TDG_PATCH_BM")) : 0;            // diagnostics: force a row tile
      const bool want128 = pbm ? pbm == 128 : c128 < c192 - 1e-9;
      if (want128 && plan_fwd_patch<128>(a, mmax)) return launch_fwd_patch<128, 208>(a, mmax, s);
      if (plan_fwd_patch<192>(a, mmax)) return launch_fwd_patch<192, 208>(a, mmax, s);
      if (!want128 && !pbm && plan_fwd_patch<128>(a, mmax)) return launch_fwd_patch<128, 208>(a, mmax, s);
    }
  }
  if (veca && bn == 208 && dma_mode && (a.N & 3) == 0 && (a.Cso & 3) == 0) {
    // one 8-wave workgroup per CU.  Row tiles: 256 (2-stage LDS ring), 192 and 128 (3-stage ring, loads two
    // steps ahead).  (Covering 400 = 208 + 192 / 800 = 2 x 208 + 2 x 192 columns exactly with a second launch of
    // 192-wide tiles -- 26 / 52 column tiles of 16 instead of 28 / 56 -- was measured: -1..-4 % on the 1536-image
    // GEMMs, +9 % on c3 backward-data and up to +70 % wherever a launch no longer fills the chip.  Not kept.)
    // Cost model for the row tile: rounds of 256 workgroups x rows per tile / measured relative efficiency.
    const long long per = (long long)tdg_ceil_div(a.N, 208) * a.nclasses;
    static const int force = getenv("TDG_DMA_BM") ? atoi(getenv("TDG_DMA_BM")) : 0;   // diagnostics
    const int bms[3] = {256, 192, 128};
    double eff[3] = {0.72, 1.0, 0.83};                // measured in the step: 800-850 / 1140 (wave-specialised) / 950-980 TF
    if (const char* e = getenv("TDG_EFF")) sscanf(e, "%lf,%lf,%lf", &eff[0], &eff[1], &eff[2]);   // diagnostics
    int best = 0;
    double best_cost = 1e30;
    for (int i = 0; i < 3; ++i) {
      const long long tiles = per * tdg_ceil_div(mmax, bms[i]);
      const double cost = (double)tdg_ceil_div(tiles, 256) * bms[i] / eff[i];
      if (cost < best_cost - 1e-9) { best_cost = cost; best = i; }
    }
    int bm = force ? force : bms[best];
    if (dma_mode == 3) bm = 192;
    if (dma_mode == 4) bm = 256;
    static const int ring192 = getenv("TDG_RING") ? atoi(getenv("TDG_RING")) : 3;     // diagnostics: 2 = 2-stage ring on the 192-row tile
    // (a 64-row tile, two workgroups per CU, was tried for the 4-step c1 forward: no change -- with K padded 75 -> 256
    //  and N 200 -> 224 that GEMM is bound by its padded MFMA work, not by serialised prologue / epilogue phases)
    if (bm == 128) {
      // a grid that leaves half of the CUs idle (the generator's 2x2 / 4x4 layers at batch 512: 64-128 workgroups)
      // takes 112-column tiles instead: twice the workgroups
      const char* n112_env = getenv("TDG_FWD_N112");            // diagnostics: 0 = never, 1 = 128-row tiles only
      if (per * tdg_ceil_div(mmax, 128) <= 128 && a.N > 112 && !(n112_env && atoi(n112_env) == 0)) {
        const long long wg112 = (long long)tdg_ceil_div(a.N, 112) * a.nclasses * tdg_ceil_div(mmax, 128);
        if (wg112 <= 128 && !(n112_env && atoi(n112_env) == 1)) return launch_fwd_dma<T, 64, 112, 3>(a, mmax, s);   // still half empty
        return launch_fwd_dma<T, 128, 112, 3>(a, mmax, s);
      }
      if constexpr (sizeof(T) == 2) {
        const char* nw_env = getenv("TDG_DMA_NW");          // diagnostics: 8 = every wave loads and computes
        const char* ws128_env = getenv("TDG_WS128");        // diagnostics: 0 = only the 192-row tile is specialised
        if (!a.accumulate && !(nw_env && atoi(nw_env) != 44) && !(ws128_env && atoi(ws128_env) == 0))
          return launch_fwd_dma<T, 128, 208, 3, 8, 1>(a, mmax, s);
      }
      return launch_fwd_dma<T, 128, 208, 3>(a, mmax, s);
    }
    // 192-row tile, bf16, staged epilogue: the wave-specialised form (4 compute + 4 loader waves; measured +5 % on
    // these launches inside the training step).  TDG_DMA_NW (diagnostics): 8 = every wave loads and computes
    // (a four-wave form, one wave per SIMD, measured -17 %, is no longer instantiated)
    const char* nw_env = getenv("TDG_DMA_NW");
    const int nw192 = nw_env ? atoi(nw_env) : 44;
    if constexpr (sizeof(T) == 2)
      if (bm == 192 && nw192 == 44 && !a.accumulate && ring192 != 2) return launch_fwd_dma<T, 192, 208, 3, 8, 1>(a, mmax, s);
    if (bm == 192) return ring192 == 2 ? launch_fwd_dma<T, 192, 208, 2>(a, mmax, s) : launch_fwd_dma<T, 192, 208, 3>(a, mmax, s);
    return launch_fwd_dma<T, 256, 208, 2>(a, mmax, s);
  }
  // 128-column problems (pix2pix / VAE widths 128, 256, 512, 1024) and 65..112 columns (the generator's 100-channel
  // layers, a 7-tile-wide column tile): the same LDS-DMA kernel, 3-stage ring at every row tile
  // 128- / 64-column problems whose K runs in 32-channel slices (pix2pix / VAE widths): 256-row tiles = four whole images or
  // one 16 x 16 block of an image, the gathered operand resident in LDS patches (igemm_fwd_patch_kernel<256, 128 | 64, 4-chunk
  // slices, 28 KiB patches>): 16 + 13 KB (128 columns) / 8 + 13 KB (64) of intake per K step instead of 16 + 32 / 8 + 16
  if constexpr (sizeof(T) == 2) {
    if (veca && dma_mode == 1 && (bn == 128 || bn == 64) && !getenv("TDG_DMA_BM") && !getenv("TDG_DMA_NW")) {
      if (bn == 128 && a.N > 112 && plan_fwd_bp<128>(a, mmax)) return launch_fwd_bp<128>(a, mmax, s);
      if (bn == 64 && a.N > 32 && plan_fwd_bp<64>(a, mmax)) return launch_fwd_bp<64>(a, mmax, s);
      if (bn == 128 && a.N > 112 && plan_fwd_patch<256, 128, 4, 28>(a, mmax)) return launch_fwd_patch<256, 128, 4, 28>(a, mmax, s);
      if (bn == 64 && a.N > 32 && plan_fwd_patch<256, 64, 4, 28>(a, mmax)) return launch_fwd_patch<256, 64, 4, 28>(a, mmax, s);
    }
  }
  if (veca && bn == 128 && dma_mode && (a.N & 3) == 0 && (a.Cso & 3) == 0) {
    const int bnt = a.N <= 112 ? 112 : 128;
    const long long per = (long long)tdg_ceil_div(a.N, bnt) * a.nclasses;
    const long long t256 = per * tdg_ceil_div(mmax, 256), t128 = per * tdg_ceil_div(mmax, 128);
    // (TWO workgroups per CU -- 128-row tiles on a 2-stage ring, 60 / 64 KB of LDS, < 90 registers -- so that one workgroup's
    //  barrier waits, prologue and epilogue lie under the other's MFMAs: measured 755 vs 827 TF (128 columns, pix2pix / VAE) and
    //  638 vs 709 TF (112 columns, the generator's dc3 at 2560 images): the smaller tile's intake per FLOP costs more than the
    //  overlap returns.  Not kept.)
    const double c256 = (double)tdg_ceil_div(t256, 256) * 256, c128 = (double)tdg_ceil_div(t128, 256) * 128 / 0.85;
    if (bnt == 112) return c128 < c256 ? launch_fwd_dma<T, 128, 112, 3>(a, mmax, s) : launch_fwd_dma<T, 256, 112, 3>(a, mmax, s);
    return c128 < c256 ? launch_fwd_dma<T, 128, 128, 3>(a, mmax, s) : launch_fwd_dma<T, 256, 128, 3>(a, mmax, s);
  }
  // 64-column problems: 128-row tile (3 loader pieces per wave on 4 column tiles)
  if (veca && bn == 64 && dma_mode && (a.N & 3) == 0 && (a.Cso & 3) == 0 && a.N > 32) return launch_fwd_dma<T, 128, 64, 3>(a, mmax, s);
  a.ntiles_n = tdg_ceil_div(a.N, bn);
  a.ntiles_m_max = tdg_ceil_div(mmax, BM);
  int gx = a.ntiles_n * a.ntiles_m_max;
  // under-filled chip (fewer than ~1.5 workgroups per CU): halve the row tile to double the grid
  if (bn == 208 && (long long)gx * a.nclasses < 384) {
    a.ntiles_m_max = tdg_ceil_div(mmax, 64);
    gx = a.ntiles_n * a.ntiles_m_max;
    return launch_fwd_cfg<T, 64, 208, 4, 1>(a, veca, gx, a.nclasses, s);
  }
  switch (bn) {
    case 16: return launch_fwd_cfg<T, BM, 16, 4, 1>(a, veca, gx, a.nclasses, s);
    case 64: return launch_fwd_cfg<T, BM, 64, 4, 1>(a, veca, gx, a.nclasses, s);
    case 128: return launch_fwd_cfg<T, BM, 128, 2, 2>(a, veca, gx, a.nclasses, s);
    case 208: return launch_fwd_cfg<T, BM, 208, 4, 1>(a, veca, gx, a.nclasses, s);
  }
  tdg_set_error("igemm_fwd: no tile config for BN=%d", bn);
  return TDG_EUNSUPPORTED;
}

template <typename T, int BKK, int BN, int WGK, int WGN>
int launch_wgrad_cfg(const WgArgs& a, bool veca, hipStream_t s) {
  constexpr int MR = WgGeom<T>::MR;
  const size_t lds = (size_t)MR * (WgGeom<T>::pitch(BKK) + WgGeom<T>::pitch(BN)) + IG_MAX_TAPS * sizeof(int);
  dim3 grid(a.ntiles_k * a.ntiles_n, 1, a.nsplit), block(256);
  static char name[64] = "";
  if (!name[0]) snprintf(name, sizeof(name), "igemm_wgrad_kernel<%s,%d,%d>", sizeof(T) == 2 ? "bf16" : "f32", BKK, BN);
  tdg_note_kernel(name);
  tdg_timing_start(name, t_flops, s);
  if (veca)
    hipLaunchKernelGGL((igemm_wgrad_kernel<T, BKK, BN, WGK, WGN, true>), grid, block, lds, s, a);
  else
    hipLaunchKernelGGL((igemm_wgrad_kernel<T, BKK, BN, WGK, WGN, false>), grid, block, lds, s, a);
  tdg_timing_stop(s);
  TDG_HIP_LAUNCH_CHECK("igemm_wgrad");
  return TDG_OK;
}

template <int BN, int MODE>
int launch_wgrad_dma(WgArgs& a, hipStream_t s) {
  const size_t lds = 4 * (size_t)WD_MR * WD_ROWB + IG_MAX_TAPS * sizeof(int);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wgrad_dma_kernel<BN, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  static char name[64] = "";
  if (!name[0]) snprintf(name, sizeof(name), "igemm_wgrad_dma_kernel<bf16,256,%d,%d>", BN, MODE);
  dim3 grid(a.ntiles_k * a.ntiles_n, 1, a.nsplit), block(512);
  tdg_note_kernel(name);
#ifdef TDG_STAMPS
  a.stamps = getenv("TDG_STAMP_PTR") ? (unsigned long long*)strtoull(getenv("TDG_STAMP_PTR"), nullptr, 0) : nullptr;
#endif
  tdg_timing_start(name, t_flops, s);
  hipLaunchKernelGGL((igemm_wgrad_dma_kernel<BN, MODE>), grid, block, lds, s, a);
  tdg_timing_stop(s);
  TDG_HIP_LAUNCH_CHECK("igemm_wgrad_dma");
  return TDG_OK;
}

template <typename T>
int launch_wgrad(WgArgs& a, bool veca, int bn, hipStream_t s) {
  switch (bn) {
    case 16: return launch_wgrad_cfg<T, 128, 16, 4, 1>(a, veca, s);
    case 64: return launch_wgrad_cfg<T, 128, 64, 4, 1>(a, veca, s);
    case 128: return launch_wgrad_cfg<T, 128, 128, 2, 2>(a, veca, s);
    case 208: return launch_wgrad_cfg<T, 128, 208, 4, 1>(a, veca, s);
  }
  tdg_set_error("igemm_wgrad: no tile config for BN=%d", bn);
  return TDG_EUNSUPPORTED;
}

inline double conv_flops(const TdgConvDesc* d, int n_images) {
  return 2.0 * n_images * d->oh * d->ow * d->kh * d->kw * (double)d->c * d->k;
}

inline short pack_tap(int dh, int dw) { return (short)((dh & 0xff) | ((dw & 0xff) << 8)); }

int validate_desc(const TdgConvDesc* d, const char* who) {
  TDG_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  TDG_CHECK_ARG(d->dtype == TDG_F32 || d->dtype == TDG_BF16, "%s: bad dtype %d", who, d->dtype);
  TDG_CHECK_ARG(d->n > 0 && d->h > 0 && d->w > 0 && d->c > 0 && d->oh > 0 && d->ow > 0 && d->k > 0, "%s: non-positive dims", who);
  TDG_CHECK_ARG(d->cs >= d->c && d->ks >= d->k, "%s: channel stride smaller than channels", who);
  TDG_CHECK_ARG(d->kh > 0 && d->kw > 0 && d->kh * d->kw <= IG_MAX_TAPS, "%s: filter %dx%d unsupported (max %d taps)", who, d->kh, d->kw, IG_MAX_TAPS);
  TDG_CHECK_ARG(d->stride >= 1 && d->stride <= 2, "%s: stride %d unsupported", who, d->stride);
  TDG_CHECK_ARG(d->pad_t >= 0 && d->pad_l >= 0 && d->pad_t < 128 && d->pad_l < 128, "%s: bad padding", who);
  // every output pixel's window must start inside the padded input
  TDG_CHECK_ARG((d->oh - 1) * d->stride - d->pad_t < d->h && (d->ow - 1) * d->stride - d->pad_l < d->w, "%s: output larger than the input admits", who);
  const long long big = (long long)d->n * d->h * d->w * d->cs * tdg_dtype_size(d->dtype);
  const long long small = (long long)d->n * d->oh * d->ow * d->ks * tdg_dtype_size(d->dtype);
  TDG_CHECK_ARG(big < 0xFFFFFF00ll && small < 0xFFFFFF00ll, "%s: tensor exceeds the 4 GiB buffer-descriptor range", who);
  return TDG_OK;
}

// effective channel count of a tensor side for the vector gather (0 -> scalar path)
// channels per tap seen by the 16-byte gather: c rounded up to a whole vector, provided the row stride
// keeps vectors aligned and has room for the round-up (the extra channels are zeros or a neighbouring
// channel window, and meet zero filter entries either way); 0 -> element-wise gather
inline int eff_channels(int c, int cs, int vec) {
  const int ce = (int)tdg_round_up(c, vec);
  return (cs % vec == 0 && ce <= cs) ? ce : 0;
}

// K order of a packed filter on the bf16 vector path: chunks (16 bytes = 8 channels) of a tap's channels per K slice.
// 5-chunk slices (C / 8 a multiple of 5: the GAN's 200 / 400 / 800 channels) are what igemm_fwd_patch_kernel keeps resident in
// LDS -- 80-byte pixels, an odd number of 16-byte chunks, so its fragment reads spread over all banks; every other
// forward-type kernel reads the sliced order through k_decode.  cv: chunks per tap.  Returns cv for the plain order.
// nrows: the GEMM's columns (rows of the packed operand): the 4-chunk order is for igemm_fwd_patch_kernel's 128- / 64-column
// tiles only (a one-column problem keeps the plain order conv_n1_fwd_kernel reads).
inline int k_slice_chunks(int dtype, int cv, int nrows) {
  static const int enabled = getenv("TDG_KSLICE") ? atoi(getenv("TDG_KSLICE")) : 1;   // diagnostics: 0 = plain order everywhere
  if (!enabled || dtype != TDG_BF16) return cv;
  if (cv >= 5 && cv % 5 == 0) return 5;
  // 32-channel slices (pix2pix / VAE widths, 64 ... 1024 channels) for the block-tile form of igemm_fwd_patch_kernel: opt-in
  // (TDG_BLOCKPATCH=1, read per call: the variant tests set it before they build their convs).  Measured in round 4 on
  // pix2pix bs 64: the block-patch kernel takes 5.87 ms where the slab kernel takes 5.34 (and the slab kernel itself loses
  // 4 % on the sliced K order): those layers run 16 - 64 K steps per workgroup and spend as long in prologue + epilogue as in
  // the loop, so less intake per step buys nothing.
  if (cv >= 8 && cv % 4 == 0 && nrows > 32 && getenv("TDG_BLOCKPATCH") && atoi(getenv("TDG_BLOCKPATCH")) == 1) return 4;
  return cv;
}

// K order of the forward filter's taps.  Stride 2: the taps of one (kh & 1, kw & 1) parity class read the same quarter
// of the input pixels (shifted by whole output pixels) and no other class touches that quarter, so walking K class by
// class keeps a workgroup's live input at 1/4 of its slab for 4-9 consecutive taps: with 32 workgroups per XCD that
// fits the 4 MB L2, where the natural row-major order re-fetched every pixel ~3x from beyond L2 (FETCH_SIZE 2.9x the
// algorithmic bytes of the 192-row kernel).  Stride 1: natural order (all taps share the region anyway).
inline int fwd_tap_order(const TdgConvDesc* d, int* ord) {
  static const int enabled = getenv("TDG_TAPORDER") ? atoi(getenv("TDG_TAPORDER")) : 1;   // diagnostics: 0 = row-major
  int n = 0;
  if (d->stride == 2 && enabled) {
    for (int pa = 0; pa < 2; ++pa)
      for (int pb = 0; pb < 2; ++pb)
        for (int kh = pa; kh < d->kh; kh += 2)
          for (int kw = pb; kw < d->kw; kw += 2) ord[n++] = kh * d->kw + kw;
  } else {
    for (int t = 0; t < d->kh * d->kw; ++t) ord[n++] = t;
  }
  return n;
}

struct BwdClassPlan {
  int ntaps, nh, nw;
  int tap_ids[IG_MAX_TAPS];
  int dh[IG_MAX_TAPS], dw[IG_MAX_TAPS];
  int GH, GW, oh0, ow0;
};

// parity classes of conv2d_backprop_input: output (big-side) pixel ih = s*a + ph takes taps
// kh == (ph + pad_t) mod s, reading small-side row a + (ph + pad_t - kh)/s.
int plan_bwd_classes(const TdgConvDesc* d, BwdClassPlan* cls) {
  const int s = d->stride;
  int nc = 0;
  for (int ph = 0; ph < s; ++ph)
    for (int pw = 0; pw < s; ++pw) {
      BwdClassPlan& c = cls[nc++];
      c.oh0 = ph;
      c.ow0 = pw;
      c.GH = (d->h - ph + s - 1) / s;
      c.GW = (d->w - pw + s - 1) / s;
      c.ntaps = c.nh = c.nw = 0;
      for (int kh = 0; kh < d->kh; ++kh) {
        if ((ph + d->pad_t - kh) % s != 0) continue;
        ++c.nh;
        c.nw = 0;
        for (int kw = 0; kw < d->kw; ++kw) {
          if ((pw + d->pad_l - kw) % s != 0) continue;
          ++c.nw;
          c.tap_ids[c.ntaps] = kh * d->kw + kw;
          c.dh[c.ntaps] = (ph + d->pad_t - kh) / s;
          c.dw[c.ntaps] = (pw + d->pad_l - kw) / s;
          ++c.ntaps;
        }
      }
    }
  return nc;
}

// ---- GEMM + col2im backward-data (bwd_col2im_kernel): when it applies and its geometry ---------------------------------
struct Col2imPlan {
  int ke, KP, NT, pitch, TA, TW, ntr, ntc, HR, HC, dh_min, dw_min, GH, GW, p_off;
  size_t w_bytes, lds;
};

bool plan_bwd_col2im(const TdgConvDesc* d, Col2imPlan* f) {
  static const int enabled = getenv("TDG_COL2IM") ? atoi(getenv("TDG_COL2IM")) : 1;   // diagnostics: 0 = fused-class / per-class kernels
  if (!enabled || d->dtype != TDG_BF16 || d->stride != 2) return false;
  if (d->kh < 2 || d->kw < 2 || d->kh > 6 || d->kw > 6) return false;
  const int ncol = d->kh * d->kw * d->c;
  static const int maxcol = getenv("TDG_C2I_MAXCOL") ? atoi(getenv("TDG_C2I_MAXCOL")) : 64;   // diagnostics (<= 80)
  if (ncol > maxcol || ncol > 80) return false;
  const int ke = eff_channels(d->k, d->ks, 8);
  if (!ke) return false;
  BwdClassPlan cls[IG_MAX_CLASSES];
  const int nc = plan_bwd_classes(d, cls);
  if (nc != 4) return false;
  int dh_lo = 1 << 20, dh_hi = -(1 << 20), dw_lo = 1 << 20, dw_hi = -(1 << 20);
  for (int i = 0; i < nc; ++i) {
    if (cls[i].ntaps > C2I_MAX_CLS_TAPS || cls[i].ntaps == 0) return false;
    for (int t = 0; t < cls[i].ntaps; ++t) {
      dh_lo = cls[i].dh[t] < dh_lo ? cls[i].dh[t] : dh_lo;
      dh_hi = cls[i].dh[t] > dh_hi ? cls[i].dh[t] : dh_hi;
      dw_lo = cls[i].dw[t] < dw_lo ? cls[i].dw[t] : dw_lo;
      dw_hi = cls[i].dw[t] > dw_hi ? cls[i].dw[t] : dw_hi;
    }
  }
  const int nhm = dh_hi - dh_lo + 1, nwm = dw_hi - dw_lo + 1;
  f->dh_min = dh_lo;
  f->dw_min = dw_lo;
  f->ke = ke;
  f->KP = (int)tdg_round_up(ke, 32);
  f->NT = tdg_ceil_div(ncol, 16);
  f->pitch = f->NT == 1 ? 16 : f->NT * 16 + 4;
  f->GH = (d->h + 1) / 2;
  f->GW = (d->w + 1) / 2;
  f->w_bytes = (size_t)f->NT * 16 * f->KP * 2;
  f->p_off = (int)tdg_round_up((long long)f->w_bytes, 256) + 256;       // filter | class tap table (256 bytes) | P
  // measured on pix2pix d8 (128 -> 1, 64 x 256 x 256): 48 KB 0.092 ms, 32 KB 0.081 ms, 16-24 KB 0.085 ms (more workgroups per CU
  // overlap the load / multiply / sum phases, smaller tiles re-read more halo); m1's input gradient (4 column tiles): 48 KB best
  static const int budget_kb = getenv("TDG_C2I_KB") ? atoi(getenv("TDG_C2I_KB")) : 0;   // diagnostics
  const size_t budget = (size_t)(budget_kb ? budget_kb : (f->NT == 1 ? 32 : 48)) * 1024;
  int ta = f->GH < 16 ? f->GH : 16, tw = f->GW < 32 ? f->GW : 32;
  auto need = [&](int a_, int w_) { return (size_t)f->p_off + (size_t)(a_ + nhm - 1) * (w_ + nwm - 1) * f->pitch * 4; };
  while (need(ta, tw) > budget && ta > 4) ta = (ta + 1) / 2;
  while (need(ta, tw) > budget && tw > 4) tw = (tw + 1) / 2;
  if (need(ta, tw) > budget) return false;
  f->ntr = tdg_ceil_div(f->GH, ta);
  f->TA = tdg_ceil_div(f->GH, f->ntr);
  f->ntc = tdg_ceil_div(f->GW, tw);
  f->TW = tdg_ceil_div(f->GW, f->ntc);
  f->HR = f->TA + nhm - 1;
  f->HC = f->TW + nwm - 1;
  if (f->HR > 255 || f->HC > 255) return false;
  f->lds = need(f->TA, f->TW);
  return true;
}

// ---- fused-class backward-data (bwd_fused_kernel): when it applies and its geometry ---------------------------
struct FusedPlan {
  int nhm, nwm, dh_min, dw_min, ntap, ke, KP, PP, wpitch, TA, ntr, TW, ntc, GH, GW, y_off;
  size_t w_bytes, lds;
};

bool plan_bwd_fused(const TdgConvDesc* d, FusedPlan* f) {
  static const int enabled = getenv("TDG_FUSE") ? atoi(getenv("TDG_FUSE")) : 1;   // diagnostics: 0 = one launch-z per class
  if (!enabled || d->dtype != TDG_BF16 || d->stride != 2 || 4 * d->c > 16) return false;
  if (d->kh < 2 || d->kw < 2) return false;                  // every class needs a tap
  Col2imPlan cp;
  if (plan_bwd_col2im(d, &cp)) return false;                 // few (tap, channel) columns: GEMM + col2im
  const int ke = eff_channels(d->k, d->ks, 8);
  if (!ke) return false;
  BwdClassPlan cls[IG_MAX_CLASSES];
  const int nc = plan_bwd_classes(d, cls);
  int dh_lo = 1 << 20, dh_hi = -(1 << 20), dw_lo = 1 << 20, dw_hi = -(1 << 20);
  for (int i = 0; i < nc; ++i)
    for (int t = 0; t < cls[i].ntaps; ++t) {
      dh_lo = cls[i].dh[t] < dh_lo ? cls[i].dh[t] : dh_lo;
      dh_hi = cls[i].dh[t] > dh_hi ? cls[i].dh[t] : dh_hi;
      dw_lo = cls[i].dw[t] < dw_lo ? cls[i].dw[t] : dw_lo;
      dw_hi = cls[i].dw[t] > dw_hi ? cls[i].dw[t] : dw_hi;
    }
  f->nhm = dh_hi - dh_lo + 1;
  f->nwm = dw_hi - dw_lo + 1;
  f->dh_min = dh_lo;
  f->dw_min = dw_lo;
  f->ntap = f->nhm * f->nwm;
  f->ke = ke;
  f->KP = (int)tdg_round_up(ke, 32);
  f->PP = ((ke / 8) & 1) ? ke : ke + 8;                       // odd number of 16-byte chunks per pixel: conflict-free fragment reads
  f->wpitch = f->ntap * f->KP + 8;                            // (ntap * KP / 8 is even)
  f->GH = (d->h + 1) / 2;
  f->GW = (d->w + 1) / 2;
  f->w_bytes = (size_t)16 * f->wpitch * 2;
  const size_t budget = 150 * 1024;
  f->y_off = (int)tdg_round_up((long long)f->w_bytes, 1024);
  const size_t fixed = (size_t)f->y_off + (size_t)f->KP * 2 + 1024;   // + slack + the tail of the halo's last wave instruction
  // Column tiles: the halo tile holds (TA + nhm - 1) x (TW + nwm - 1) small-side pixels.  With full-width rows a wide,
  // many-channel small side (pix2pix d8: 128 anchors x 128 channels) fits ONE anchor row, i.e. every source row is staged
  // nhm = 3 times; the fewest column tiles whose halo re-read factor (rows and columns) is within 10 % of the best wins.
  int best_ntc = 0, best_ta = 0;
  double best_amp = 1e30;
  for (int ntc = 1; ntc <= 8; ++ntc) {
    const int tw = tdg_ceil_div(f->GW, ntc);
    const size_t row_bytes = (size_t)(tw + f->nwm - 1) * f->PP * 2;
    if (fixed + (size_t)f->nhm * row_bytes > budget) continue;
    int ta = (int)((budget - fixed) / row_bytes) - (f->nhm - 1);
    if (ta > f->GH) ta = f->GH;
    if (ta < 1) continue;
    const int ntr = tdg_ceil_div(f->GH, ta);
    ta = tdg_ceil_div(f->GH, ntr);                            // balanced row tiles
    const double amp = (double)(ta + f->nhm - 1) / ta * (double)(tw + f->nwm - 1) / tw;
    if (amp < best_amp * 0.9) { best_amp = amp; best_ntc = ntc; best_ta = ta; }
  }
  if (!best_ntc) return false;
  f->ntc = best_ntc;
  f->TW = tdg_ceil_div(f->GW, best_ntc);
  f->TA = best_ta;
  f->ntr = tdg_ceil_div(f->GH, best_ta);
  f->lds = fixed + (size_t)(f->TA + f->nhm - 1) * (size_t)(f->TW + f->nwm - 1) * f->PP * 2;
  return true;
}

// ---- thin-input forward conv (thin_fwd_kernel): when it applies and its geometry ------------------------------------
struct ThinPlan {
  int NC;                 // output columns per tile: 208 or 64
  int Kp, WP, TH, tiles_per_image, ntiles_n, PH, PW, y_off, k_off, rows;
  size_t w_bytes, lds;
};

bool plan_fwd_thin(const TdgConvDesc* d, ThinPlan* t) {
  static const int enabled = getenv("TDG_THIN") ? atoi(getenv("TDG_THIN")) : 1;   // diagnostics: 0 = implicit-GEMM kernels, 2 = only the 208-column form
  if (!enabled || d->dtype != TDG_BF16 || d->c > 4 || d->cs < 4 || (d->cs & 3) || (d->k & 7) || (d->ks & 7)) return false;
  if (d->ow > TH_PIX) return false;
  // 208-column tiles for wide outputs; 64-column tiles for 32 .. 128 output channels when the descriptor's batch has >= 64k
  // output pixels (smaller launches stay with the implicit-GEMM kernels: too short for the difference to show)
  if (d->k >= 160) t->NC = 208;
  else if (enabled != 2 && d->k >= 32 && d->k <= 128 && (long long)d->n * d->oh * d->ow >= 65536) t->NC = 64;
  else return false;
  const int tile_rows = t->NC == 208 ? 224 : 64;
  t->Kp = (int)tdg_round_up((long long)d->kh * d->kw * d->c, 32);
  if (t->Kp > 256) return false;
  t->WP = ((t->Kp / 8) & 1) ? t->Kp : t->Kp + 8;               // odd number of 16-byte chunks per filter row
  t->TH = TH_PIX / d->ow;
  if (t->TH > d->oh) t->TH = d->oh;
  t->tiles_per_image = tdg_ceil_div(d->oh, t->TH);
  t->ntiles_n = tdg_ceil_div(d->k, t->NC);
  t->PH = (t->TH - 1) * d->stride + d->kh;
  t->PW = (d->ow - 1) * d->stride + d->kw;
  if ((long long)t->PH * t->PW * 4 + 8 > 0xfff0) return false; // 16-bit patch offsets
  t->rows = t->ntiles_n * tile_rows;
  t->w_bytes = (size_t)t->rows * t->WP * 2;
  t->y_off = (int)tdg_round_up((long long)tile_rows * t->WP * 2, 1024);         // + the zero-filled tail of the last filter DMA
  t->k_off = t->y_off + (int)tdg_round_up(((long long)t->PH * t->PW * 2 + 4) * 4, 256);   // + the tail of the last patch DMA
  const size_t main = (size_t)t->k_off + (size_t)t->Kp * 2 + 16;
  const size_t epi = (size_t)TH_PIX * (t->NC * 2 + 16);
  t->lds = main > epi ? main : epi;
  return t->lds <= 80 * 1024;                                  // two workgroups per CU
}

}  // namespace

extern "C" {

size_t tdg_packed_filter_fwd_bytes(const TdgConvDesc* d) {
  if (validate_desc(d, "tdg_packed_filter_fwd_bytes") != TDG_OK) return 0;
  ThinPlan tp;
  if (plan_fwd_thin(d, &tp)) return tp.w_bytes;
  const int es = tdg_dtype_size(d->dtype), vec = 16 / es, bke = IG_BKB / es;
  int ce = eff_channels(d->c, d->cs, vec);
  if (!ce) ce = d->c;
  const long long K = (long long)d->kh * d->kw * ce;
  return (size_t)d->k * (size_t)tdg_round_up(K, bke) * es;
}

size_t tdg_packed_filter_bwd_bytes(const TdgConvDesc* d) {
  if (validate_desc(d, "tdg_packed_filter_bwd_bytes") != TDG_OK) return 0;
  const int es = tdg_dtype_size(d->dtype), vec = 16 / es, bke = IG_BKB / es;
  int ke = eff_channels(d->k, d->ks, vec);
  if (!ke) ke = d->k;
  Col2imPlan cp;
  if (plan_bwd_col2im(d, &cp)) return cp.w_bytes;
  FusedPlan fp;
  if (plan_bwd_fused(d, &fp)) return fp.w_bytes;
  BwdClassPlan cls[IG_MAX_CLASSES];
  const int nc = plan_bwd_classes(d, cls);
  size_t total = 0;
  for (int i = 0; i < nc; ++i) total += (size_t)d->c * (size_t)tdg_round_up((long long)cls[i].ntaps * ke, bke) * es;
  return total;
}

// ---- packing jobs ------------------------------------------------------------------------------------------------
static void build_pack_fwd(const TdgConvDesc* d, const float* w, void* packed, PackArgs* a) {
  const int es = tdg_dtype_size(d->dtype), vec = 16 / es, bke = IG_BKB / es;
  int ce = eff_channels(d->c, d->cs, vec);
  if (!ce) ce = d->c;
  a->w = w;
  a->out = packed;
  a->rows = d->k;
  a->ntaps = d->kh * d->kw;
  a->C = d->c;
  a->Ceff = ce;
  a->CK = (ce % vec == 0 && eff_channels(d->c, d->cs, vec)) ? k_slice_chunks(d->dtype, ce / vec, d->k) * vec : ce;
  a->Kp = (int)tdg_round_up((long long)a->ntaps * ce, bke);
  a->stride_tap = d->c * d->k;
  a->stride_row = 1;       // row = small-side channel (last master index)
  a->stride_ch = d->k;     // K channel = big-side channel
  int ord[IG_MAX_TAPS];
  fwd_tap_order(d, ord);
  for (int t = 0; t < a->ntaps; ++t) a->tap_ids[t] = (unsigned char)ord[t];
  pack_finish(a);
}

// one job per non-empty parity class; returns the count (the fused-class form is not handled here)
static int build_pack_bwd(const TdgConvDesc* d, const float* w, void* packed, PackArgs* out) {
  const int es = tdg_dtype_size(d->dtype), vec = 16 / es, bke = IG_BKB / es;
  int ke = eff_channels(d->k, d->ks, vec);
  if (!ke) ke = d->k;
  BwdClassPlan cls[IG_MAX_CLASSES];
  const int nc = plan_bwd_classes(d, cls);
  size_t off = 0;
  int n = 0;
  for (int i = 0; i < nc; ++i) {
    if (cls[i].ntaps == 0) continue;
    PackArgs& a = out[n++];
    a.w = w;
    a.out = static_cast<char*>(packed) + off;
    a.rows = d->c;
    a.ntaps = cls[i].ntaps;
    a.C = d->k;
    a.Ceff = ke;
    a.CK = (ke % vec == 0 && eff_channels(d->k, d->ks, vec)) ? k_slice_chunks(d->dtype, ke / vec, d->c) * vec : ke;
    a.Kp = (int)tdg_round_up((long long)a.ntaps * ke, bke);
    a.stride_tap = d->c * d->k;
    a.stride_row = d->k;    // row = big-side channel
    a.stride_ch = 1;        // K channel = small-side channel (contiguous in the master)
    for (int t = 0; t < a.ntaps; ++t) a.tap_ids[t] = (unsigned char)cls[i].tap_ids[t];
    pack_finish(&a);
    off += (size_t)a.rows * a.Kp * es;
  }
  return n;
}

// bwd_col2im_kernel's filter [tap * C + c][KP] is the master [kh][kw][c][k] cast row by row
static void build_pack_col2im(const TdgConvDesc* d, const Col2imPlan& cp, const float* w, void* packed, PackArgs* a) {
  a->w = w;
  a->out = packed;
  a->rows = d->kh * d->kw * d->c;
  a->ntaps = 1;
  a->C = d->k;
  a->Ceff = cp.ke;
  a->CK = cp.ke;
  a->Kp = cp.KP;
  a->stride_tap = 0;
  a->stride_row = d->k;
  a->stride_ch = 1;
  a->tap_ids[0] = 0;
  pack_finish(a);
}

static int launch_pack_one(const PackArgs& a, int dtype, hipStream_t s) {
  dim3 grid(tdg_ceil_div(a.Kp, PACK_TK), tdg_ceil_div(a.rows, 32 * PACK_RT));
  if (dtype == TDG_BF16)
    hipLaunchKernelGGL(pack_filter_kernel<bf16_t>, grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(pack_filter_kernel<float>, grid, dim3(256), 0, s, a);
  TDG_HIP_LAUNCH_CHECK("pack_filter");
  return TDG_OK;
}

static void fill_pack_fused(const TdgConvDesc* d, const FusedPlan& fp, const float* w, void* packed, FusedPackArgs* a) {
  a->w = w;
  a->out = static_cast<bf16_t*>(packed);
  a->C = d->c; a->K = d->k; a->KH = d->kh; a->KW = d->kw; a->pad_t = d->pad_t; a->pad_l = d->pad_l;
  a->nhm = fp.nhm; a->nwm = fp.nwm; a->dh_min = fp.dh_min; a->dw_min = fp.dw_min; a->KP = fp.KP; a->wpitch = fp.wpitch;
}
static int launch_pack_fused(const TdgConvDesc* d, const FusedPlan& fp, const float* w, void* packed, hipStream_t s) {
  FusedPackArgs a;
  fill_pack_fused(d, fp, w, packed, &a);
  hipLaunchKernelGGL(pack_fused_kernel, dim3(tdg_ceil_div(16 * fp.wpitch, 256)), dim3(256), 0, s, a);
  TDG_HIP_LAUNCH_CHECK("pack_filter_bwd(fused)");
  return TDG_OK;
}

static void fill_pack_thin(const TdgConvDesc* d, const ThinPlan& tp, const float* w, void* packed, ThinPackArgs* a) {
  a->w = w;
  a->out = static_cast<bf16_t*>(packed);
  a->C = d->c; a->N = d->k; a->taps = d->kh * d->kw; a->Kp = tp.Kp; a->WP = tp.WP; a->rows = tp.rows;
  a->tile_rows = tp.NC == 208 ? 224 : 64; a->tile_cols = tp.NC;
}
static int launch_pack_thin(const TdgConvDesc* d, const ThinPlan& tp, const float* w, void* packed, hipStream_t s) {
  ThinPackArgs a;
  fill_pack_thin(d, tp, w, packed, &a);
  hipLaunchKernelGGL(pack_thin_kernel, dim3(tdg_ceil_div((long long)tp.rows * tp.WP, 256)), dim3(256), 0, s, a);
  TDG_HIP_LAUNCH_CHECK("pack_filter_fwd(thin)");
  return TDG_OK;
}

int tdg_pack_filter_fwd(const TdgConvDesc* d, const float* w, void* packed, void* stream) {
  int rc = validate_desc(d, "tdg_pack_filter_fwd");
  if (rc) return rc;
  TDG_CHECK_ARG(w && packed, "tdg_pack_filter_fwd: null pointer");
  ThinPlan tp;
  if (plan_fwd_thin(d, &tp)) return launch_pack_thin(d, tp, w, packed, (hipStream_t)stream);
  PackArgs a;
  build_pack_fwd(d, w, packed, &a);
  return launch_pack_one(a, d->dtype, (hipStream_t)stream);
}

int tdg_pack_filter_bwd(const TdgConvDesc* d, const float* w, void* packed, void* stream) {
  int rc = validate_desc(d, "tdg_pack_filter_bwd");
  if (rc) return rc;
  TDG_CHECK_ARG(w && packed, "tdg_pack_filter_bwd: null pointer");
  Col2imPlan cp;
  if (plan_bwd_col2im(d, &cp)) {
    PackArgs a;
    build_pack_col2im(d, cp, w, packed, &a);
    return launch_pack_one(a, d->dtype, (hipStream_t)stream);
  }
  FusedPlan fp;
  if (plan_bwd_fused(d, &fp)) return launch_pack_fused(d, fp, w, packed, (hipStream_t)stream);
  PackArgs a[IG_MAX_CLASSES];
  const int n = build_pack_bwd(d, w, packed, a);
  for (int i = 0; i < n; ++i) {
    rc = launch_pack_one(a[i], d->dtype, (hipStream_t)stream);
    if (rc) return rc;
  }
  return TDG_OK;
}

int tdg_pack_filters(const TdgPackJob* jobs, int n_jobs, void* stream) {
  TDG_CHECK_ARG(jobs && n_jobs > 0, "tdg_pack_filters: no jobs");
  const int dtype = jobs[0].desc.dtype;
  static PackMultiArgs m;                       // 4 KB: built on the host, passed by value at launch (one host thread per GPU)
  m.njobs = 0;
  m.start[0] = 0;
  m.thin_blocks = m.fused_blocks = 0;
  auto flush = [&]() -> int {
    const int nb = m.start[m.njobs] + m.thin_blocks + m.fused_blocks;
    if (nb == 0) return TDG_OK;
    if (dtype == TDG_BF16)
      hipLaunchKernelGGL(pack_multi_kernel<bf16_t>, dim3(nb), dim3(256), 0, (hipStream_t)stream, m);
    else
      hipLaunchKernelGGL(pack_multi_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, m);
    TDG_HIP_LAUNCH_CHECK("pack_filters");
    m.njobs = 0;
    m.thin_blocks = m.fused_blocks = 0;
    return TDG_OK;
  };
  auto push = [&](const PackArgs& a) -> int {
    if (m.njobs == PACK_MULTI_MAX) {
      const int rc = flush();
      if (rc) return rc;
    }
    m.job[m.njobs] = a;
    m.start[m.njobs + 1] = m.start[m.njobs] + tdg_ceil_div(a.Kp, PACK_TK) * tdg_ceil_div(a.rows, 32 * PACK_RT);
    ++m.njobs;
    return TDG_OK;
  };
  for (int j = 0; j < n_jobs; ++j) {
    const TdgConvDesc* d = &jobs[j].desc;
    int rc = validate_desc(d, "tdg_pack_filters");
    if (rc) return rc;
    TDG_CHECK_ARG(d->dtype == dtype, "tdg_pack_filters: job %d has dtype %d, job 0 has %d", j, d->dtype, dtype);
    TDG_CHECK_ARG(jobs[j].w, "tdg_pack_filters: job %d has no master filter", j);
    if (jobs[j].packed_fwd) {
      ThinPlan tp;
      if (plan_fwd_thin(d, &tp)) {
        if (m.thin_blocks == 0) {                              // rides in the multi-job launch (the first one of a call)
          fill_pack_thin(d, tp, jobs[j].w, jobs[j].packed_fwd, &m.thin);
          m.thin_blocks = tdg_ceil_div((long long)tp.rows * tp.WP, 256);
        } else {
          rc = launch_pack_thin(d, tp, jobs[j].w, jobs[j].packed_fwd, (hipStream_t)stream);
          if (rc) return rc;
        }
      } else {
        PackArgs a;
        build_pack_fwd(d, jobs[j].w, jobs[j].packed_fwd, &a);
        rc = push(a);
        if (rc) return rc;
      }
    }
    if (jobs[j].packed_bwd) {
      FusedPlan fp;
      Col2imPlan cp;
      if (plan_bwd_col2im(d, &cp)) {
        PackArgs a;
        build_pack_col2im(d, cp, jobs[j].w, jobs[j].packed_bwd, &a);
        rc = push(a);
        if (rc) return rc;
      } else if (plan_bwd_fused(d, &fp)) {
        if (m.fused_blocks == 0) {
          fill_pack_fused(d, fp, jobs[j].w, jobs[j].packed_bwd, &m.fused);
          m.fused_blocks = tdg_ceil_div(16 * fp.wpitch, 256);
        } else {
          rc = launch_pack_fused(d, fp, jobs[j].w, jobs[j].packed_bwd, (hipStream_t)stream);
          if (rc) return rc;
        }
      } else {
        PackArgs a[IG_MAX_CLASSES];
        const int n = build_pack_bwd(d, jobs[j].w, jobs[j].packed_bwd, a);
        for (int i = 0; i < n; ++i) {
          rc = push(a[i]);
          if (rc) return rc;
        }
      }
    }
  }
  return flush();
}

static void fill_epilogue(IgArgs& a, const TdgEpilogue* epi) {
  a.bias = epi ? epi->bias : nullptr;
  a.act = epi ? epi->act : TDG_ACT_NONE;
  a.leak = epi ? epi->leak : 0.f;
  a.mask_mode = epi ? epi->mask_mode : TDG_MASK_NONE;
  a.mask_src = epi ? epi->mask_src : nullptr;
  a.accumulate = epi ? epi->accumulate : 0;
  if (a.mask_mode == TDG_MASK_NONE) a.mask_src = nullptr;
  // column partials: granted by launch_fwd_dma when the chosen kernel has the staged bf16 epilogue, else reported as 0 tiles
  a.col_partial = nullptr;
  a.col_mode = TDG_COL_NONE;
  a.col_images = 0;
  a.ksplit = 1;
  a.steps_per_split = 1 << 30;
  a.slab = nullptr;
  t_splitk = (epi && epi->splitk_ws && epi->splitk_ws_bytes) ? epi : nullptr;
  t_col = nullptr;
  if (epi && epi->col_partial && epi->col_mode != TDG_COL_NONE && epi->col_nblk_out) {
    *epi->col_nblk_out = 0;
    t_col = epi;
  }
}

int tdg_conv2d_fwd(const TdgConvDesc* d, int n_images, const void* x, const void* wp, void* y,
                   const TdgEpilogue* epi, void* stream) {
  int rc = validate_desc(d, "tdg_conv2d_fwd");
  if (rc) return rc;
  TDG_CHECK_ARG(n_images > 0 && n_images <= d->n, "tdg_conv2d_fwd: n_images %d outside (0, %d]", n_images, d->n);
  TDG_CHECK_ARG(x && wp && y, "tdg_conv2d_fwd: null pointer");
  TDG_CHECK_ARG(!epi || epi->mask_mode == TDG_MASK_NONE || epi->mask_src, "tdg_conv2d_fwd: mask without mask_src");
  const int es = tdg_dtype_size(d->dtype), vec = 16 / es, bke = IG_BKB / es;
  const int ce = eff_channels(d->c, d->cs, vec);
  const bool veca = ce != 0;
  const int C = veca ? ce : d->c;
  TDG_CHECK_ARG(!veca || ((uintptr_t)x & 15) == 0, "tdg_conv2d_fwd: x must be 16-byte aligned (channel stride allows the vector gather)");
  ThinPlan tp;
  if (plan_fwd_thin(d, &tp) && !(epi && epi->accumulate)) {
    ThinFwdArgs f;
    memset(&f, 0, sizeof(f));
    f.x = static_cast<const bf16_t*>(x);
    f.w = static_cast<const bf16_t*>(wp);
    f.y = static_cast<bf16_t*>(y);
    f.bias = epi ? epi->bias : nullptr;
    f.act = epi ? epi->act : TDG_ACT_NONE;
    f.leak = epi ? epi->leak : 0.f;
    f.mask_mode = epi ? epi->mask_mode : TDG_MASK_NONE;
    f.mask_src = f.mask_mode != TDG_MASK_NONE ? static_cast<const bf16_t*>(epi->mask_src) : nullptr;
    f.H = d->h; f.W = d->w; f.Cs = d->cs; f.C = d->c;
    f.OH = d->oh; f.OW = d->ow; f.Cso = d->ks; f.N = d->k;
    f.KH = d->kh; f.KW = d->kw; f.stride = d->stride; f.pad_t = d->pad_t; f.pad_l = d->pad_l;
    f.Kp = tp.Kp; f.WP = tp.WP; f.TH = tp.TH; f.tiles_per_image = tp.tiles_per_image; f.ntiles_n = tp.ntiles_n;
    f.PH = tp.PH; f.PW = tp.PW; f.y_off = tp.y_off; f.k_off = tp.k_off;
    f.fd_pw = make_fastdiv(tp.PW);
    f.fd_ow = make_fastdiv(d->ow);
    f.fd_c = make_fastdiv(d->c);
    f.debug = 0;
#ifdef TDG_STAMPS
    f.debug = getenv("TDG_DEBUG_ABLATE") ? atoi(getenv("TDG_DEBUG_ABLATE")) : 0;     // (garbage results: diagnostic library only)
#endif
    f.stamps = nullptr;
#ifdef TDG_STAMPS
    f.stamps = getenv("TDG_STAMP_PTR") ? (unsigned long long*)strtoull(getenv("TDG_STAMP_PTR"), nullptr, 0) : nullptr;
#endif
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_fwd_kernel<208>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_fwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      attr_set = true;
    }
    const char* kname = tp.NC == 208 ? "thin_fwd_kernel<bf16>" : "thin_fwd_kernel<bf16,64>";
    tdg_note_kernel(kname);
    tdg_timing_start(kname, conv_flops(d, n_images), (hipStream_t)stream);
    const dim3 tgrid(n_images * tp.tiles_per_image * tp.ntiles_n);
    if (tp.NC == 208)
      hipLaunchKernelGGL(thin_fwd_kernel<208>, tgrid, dim3(256), tp.lds, (hipStream_t)stream, f);
    else
      hipLaunchKernelGGL(thin_fwd_kernel<64>, tgrid, dim3(256), tp.lds, (hipStream_t)stream, f);
    tdg_timing_stop((hipStream_t)stream);
    TDG_HIP_LAUNCH_CHECK("thin_fwd");
    return TDG_OK;
  }
  static const int n1_enabled = getenv("TDG_CONVN1") ? atoi(getenv("TDG_CONVN1")) : 1;   // diagnostics: 0 = the 128 x 16 MFMA tile
  if (n1_enabled && d->k == 1 && veca && k_slice_chunks(d->dtype, C / vec, d->k) == C / vec && d->kh * d->kw <= IG_MAX_TAPS && (long long)n_images * d->oh * d->ow >= 1024 &&
      !(epi && (epi->accumulate || epi->mask_mode != TDG_MASK_NONE))) {
    if (epi && epi->col_nblk_out) *epi->col_nblk_out = 0;       // no column partials from this kernel: the host runs its own pass
    ConvN1Args f;
    memset(&f, 0, sizeof(f));
    f.x = x; f.w = wp; f.y = y;
    f.bias = epi ? epi->bias : nullptr;
    f.act = epi ? epi->act : TDG_ACT_NONE;
    f.leak = epi ? epi->leak : 0.f;
    f.x_bytes = (unsigned)((long long)n_images * d->h * d->w * d->cs * es);
    f.w_bytes = (unsigned)tdg_packed_filter_fwd_bytes(d);
    f.SH = d->h; f.SW = d->w; f.Cs = d->cs; f.C = C; f.OH = d->oh; f.OW = d->ow; f.Cso = d->ks; f.stride = d->stride;
    f.ntaps = d->kh * d->kw;
    f.M = n_images * d->oh * d->ow;
    int ord[IG_MAX_TAPS];
    fwd_tap_order(d, ord);
    for (int t = 0; t < f.ntaps; ++t) f.tap[t] = pack_tap(ord[t] / d->kw - d->pad_t, ord[t] % d->kw - d->pad_l);
    f.fd_ohw = make_fastdiv(d->oh * d->ow);
    f.fd_ow = make_fastdiv(d->ow);
    const char* name = d->dtype == TDG_BF16 ? "conv_n1_fwd_kernel<bf16>" : "conv_n1_fwd_kernel<f32>";
    tdg_note_kernel(name);
    tdg_timing_start(name, conv_flops(d, n_images), (hipStream_t)stream);
    if (d->dtype == TDG_BF16)
      hipLaunchKernelGGL(conv_n1_fwd_kernel<bf16_t>, dim3(tdg_ceil_div(f.M, 4)), dim3(256), 0, (hipStream_t)stream, f);
    else
      hipLaunchKernelGGL(conv_n1_fwd_kernel<float>, dim3(tdg_ceil_div(f.M, 4)), dim3(256), 0, (hipStream_t)stream, f);
    tdg_timing_stop((hipStream_t)stream);
    TDG_HIP_LAUNCH_CHECK("conv_n1_fwd");
    return TDG_OK;
  }
  IgArgs a;
  memset(&a, 0, sizeof(a));
  a.src = x;
  a.wpack = wp;
  a.out = y;
  fill_epilogue(a, epi);
  a.src_bytes = (unsigned)((long long)n_images * d->h * d->w * d->cs * es);
  a.w_bytes = (unsigned)tdg_packed_filter_fwd_bytes(d);
  a.SH = d->h; a.SW = d->w; a.sigma = d->stride;
  a.C = C; a.Cs = d->cs;
  a.fd_c = make_fastdiv(veca ? C / vec : C);
  a.fd_ck = make_fastdiv(veca ? k_slice_chunks(d->dtype, C / vec, d->k) : C);
  a.nslices = (int)(a.fd_c.d / a.fd_ck.d);
  a.N = d->k; a.OH = d->oh; a.OW = d->ow; a.os = 1; a.Cso = d->ks;
  a.nclasses = 1;
  IgClass& c = a.cls[0];
  c.M = n_images * d->oh * d->ow;
  c.GH = d->oh; c.GW = d->ow;
  c.ntaps = d->kh * d->kw;
  c.K = c.ntaps * C;
  c.nsteps = tdg_ceil_div(c.K, bke);
  c.Kp = c.nsteps * bke;
  c.oh0 = c.ow0 = 0;
  c.w_off_bytes = 0;
  c.fd_ghw = make_fastdiv(c.GH * c.GW);
  c.fd_gw = make_fastdiv(c.GW);
  int ord[IG_MAX_TAPS];
  fwd_tap_order(d, ord);
  for (int t = 0; t < c.ntaps; ++t) c.tap[t] = pack_tap(ord[t] / d->kw - d->pad_t, ord[t] % d->kw - d->pad_l);
  c.nh = d->kh; c.nw = d->kw; c.dh0 = -d->pad_t; c.dw0 = -d->pad_l; c.sh = c.sw = 1;
  c.fd_nw = make_fastdiv(c.nw);
  c.fd_nt = make_fastdiv(c.ntaps);
  const int bn = pick_bn(d->k);
  t_flops = conv_flops(d, n_images);
  return d->dtype == TDG_BF16 ? launch_fwd<bf16_t>(a, veca, bn, (hipStream_t)stream)
                              : launch_fwd<float>(a, veca, bn, (hipStream_t)stream);
}

int tdg_conv2d_bwd_data(const TdgConvDesc* d, int n_images, const void* y, const void* wp, void* x,
                        const TdgEpilogue* epi, void* stream) {
  int rc = validate_desc(d, "tdg_conv2d_bwd_data");
  if (rc) return rc;
  TDG_CHECK_ARG(n_images > 0 && n_images <= d->n, "tdg_conv2d_bwd_data: n_images %d outside (0, %d]", n_images, d->n);
  TDG_CHECK_ARG(x && wp && y, "tdg_conv2d_bwd_data: null pointer");
  TDG_CHECK_ARG(!epi || epi->mask_mode == TDG_MASK_NONE || epi->mask_src, "tdg_conv2d_bwd_data: mask without mask_src");
  const int es = tdg_dtype_size(d->dtype), vec = 16 / es, bke = IG_BKB / es;
  const int ke = eff_channels(d->k, d->ks, vec);
  const bool veca = ke != 0;
  const int C = veca ? ke : d->k;
  TDG_CHECK_ARG(!veca || ((uintptr_t)y & 15) == 0, "tdg_conv2d_bwd_data: y must be 16-byte aligned (channel stride allows the vector gather)");
  Col2imPlan cp;
  if (plan_bwd_col2im(d, &cp)) {
    Col2imArgs f;
    memset(&f, 0, sizeof(f));
    f.y = static_cast<const bf16_t*>(y);
    f.w = static_cast<const bf16_t*>(wp);
    f.x = static_cast<bf16_t*>(x);
    f.bias = epi ? epi->bias : nullptr;
    f.act = epi ? epi->act : TDG_ACT_NONE;
    f.leak = epi ? epi->leak : 0.f;
    f.mask_mode = epi ? epi->mask_mode : TDG_MASK_NONE;
    f.mask_src = f.mask_mode != TDG_MASK_NONE ? static_cast<const bf16_t*>(epi->mask_src) : nullptr;
    f.accumulate = epi ? epi->accumulate : 0;
    f.SH = d->oh; f.SW = d->ow; f.Cs = d->ks;
    f.OH = d->h; f.OW = d->w; f.Cso = d->cs; f.C = d->c;
    f.KP = cp.KP; f.NT = cp.NT; f.pitch = cp.pitch; f.ke = cp.ke;
    f.TA = cp.TA; f.TW = cp.TW; f.ntr = cp.ntr; f.ntc = cp.ntc; f.HR = cp.HR; f.HC = cp.HC; f.dh_min = cp.dh_min; f.dw_min = cp.dw_min;
    f.p_off = cp.p_off;
    f.debug = 0;
#ifdef TDG_STAMPS
    f.debug = getenv("TDG_DEBUG_ABLATE") ? atoi(getenv("TDG_DEBUG_ABLATE")) : 0;     // (garbage results: diagnostic library only)
#endif
    BwdClassPlan plan[IG_MAX_CLASSES];
    plan_bwd_classes(d, plan);
    for (int i = 0; i < 4; ++i) {
      const int ci = 2 * plan[i].oh0 + plan[i].ow0;
      f.cls_ntaps[ci] = plan[i].ntaps;
      for (int t = 0; t < plan[i].ntaps; ++t)
        f.cls_tap[ci][t] = ((plan[i].dh[t] - cp.dh_min) << 16) | ((plan[i].dw[t] - cp.dw_min) << 8) | plan[i].tap_ids[t];
    }
    f.fd_hc = make_fastdiv(cp.HC);
    f.fd_c = make_fastdiv(d->c);
    f.fd_ow2 = make_fastdiv(2 * cp.TW);
    tdg_note_kernel("bwd_col2im_kernel<bf16>");
    tdg_timing_start("bwd_col2im_kernel<bf16>", conv_flops(d, n_images), (hipStream_t)stream);
    const dim3 grid(n_images * cp.ntr * cp.ntc);
    switch (cp.NT) {
      case 1: hipLaunchKernelGGL(bwd_col2im_kernel<1>, grid, dim3(256), cp.lds, (hipStream_t)stream, f); break;
      case 2: hipLaunchKernelGGL(bwd_col2im_kernel<2>, grid, dim3(256), cp.lds, (hipStream_t)stream, f); break;
      case 3: hipLaunchKernelGGL(bwd_col2im_kernel<3>, grid, dim3(256), cp.lds, (hipStream_t)stream, f); break;
      case 4: hipLaunchKernelGGL(bwd_col2im_kernel<4>, grid, dim3(256), cp.lds, (hipStream_t)stream, f); break;
      default: hipLaunchKernelGGL(bwd_col2im_kernel<5>, grid, dim3(256), cp.lds, (hipStream_t)stream, f); break;
    }
    tdg_timing_stop((hipStream_t)stream);
    TDG_HIP_LAUNCH_CHECK("bwd_col2im");
    return TDG_OK;
  }
  FusedPlan fp;
  if (plan_bwd_fused(d, &fp)) {
    FusedBwdArgs f;
    memset(&f, 0, sizeof(f));
    f.y = static_cast<const bf16_t*>(y);
    f.w = static_cast<const bf16_t*>(wp);
    f.x = static_cast<bf16_t*>(x);
    f.bias = epi ? epi->bias : nullptr;
    f.act = epi ? epi->act : TDG_ACT_NONE;
    f.leak = epi ? epi->leak : 0.f;
    f.mask_mode = epi ? epi->mask_mode : TDG_MASK_NONE;
    f.mask_src = f.mask_mode != TDG_MASK_NONE ? static_cast<const bf16_t*>(epi->mask_src) : nullptr;
    f.accumulate = epi ? epi->accumulate : 0;
    f.SH = d->oh; f.SW = d->ow; f.Cs = d->ks; f.ke = fp.ke;
    f.OH = d->h; f.OW = d->w; f.Cso = d->cs; f.C = d->c;
    f.GH = fp.GH; f.GW = fp.GW;
    f.nhm = fp.nhm; f.nwm = fp.nwm; f.dh_min = fp.dh_min; f.dw_min = fp.dw_min; f.ntap = fp.ntap;
    f.KP = fp.KP; f.PP = fp.PP; f.wpitch = fp.wpitch; f.TA = fp.TA; f.ntr = fp.ntr; f.TW = fp.TW; f.ntc = fp.ntc; f.y_off = fp.y_off;
    f.fd_vpp = make_fastdiv(fp.PP / 8);
    f.fd_hc = make_fastdiv(fp.TW + fp.nwm - 1);
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_set = true;
    }
#ifdef TDG_STAMPS
    f.stamps = getenv("TDG_STAMP_PTR") ? (unsigned long long*)strtoull(getenv("TDG_STAMP_PTR"), nullptr, 0) : nullptr;
#endif
    tdg_note_kernel("bwd_fused_kernel<bf16>");
    tdg_timing_start("bwd_fused_kernel<bf16>", conv_flops(d, n_images), (hipStream_t)stream);
    hipLaunchKernelGGL(bwd_fused_kernel, dim3(n_images * fp.ntr * fp.ntc), dim3(512), fp.lds, (hipStream_t)stream, f);
    tdg_timing_stop((hipStream_t)stream);
    TDG_HIP_LAUNCH_CHECK("bwd_fused");
    return TDG_OK;
  }
  IgArgs a;
  memset(&a, 0, sizeof(a));
  a.src = y;
  a.wpack = wp;
  a.out = x;
  fill_epilogue(a, epi);
  a.src_bytes = (unsigned)((long long)n_images * d->oh * d->ow * d->ks * es);
  a.w_bytes = (unsigned)tdg_packed_filter_bwd_bytes(d);
  a.SH = d->oh; a.SW = d->ow; a.sigma = 1;
  a.C = C; a.Cs = d->ks;
  a.fd_c = make_fastdiv(veca ? C / vec : C);
  a.fd_ck = make_fastdiv(veca ? k_slice_chunks(d->dtype, C / vec, d->c) : C);
  a.nslices = (int)(a.fd_c.d / a.fd_ck.d);
  a.N = d->c; a.OH = d->h; a.OW = d->w; a.os = d->stride; a.Cso = d->cs;
  BwdClassPlan plan[IG_MAX_CLASSES];
  const int nc = plan_bwd_classes(d, plan);
  a.nclasses = 0;
  unsigned off = 0;
  for (int i = 0; i < nc; ++i) {
    const int Kp = (int)tdg_round_up((long long)plan[i].ntaps * C, bke);
    if (plan[i].ntaps == 0 || plan[i].GH <= 0 || plan[i].GW <= 0) {
      // a class without taps would leave its output pixels unwritten: only possible when the
      // filter is smaller than the stride, which the reference never uses
      TDG_CHECK_ARG(plan[i].GH <= 0 || plan[i].GW <= 0, "tdg_conv2d_bwd_data: filter smaller than stride");
      continue;
    }
    IgClass& c = a.cls[a.nclasses++];
    c.M = n_images * plan[i].GH * plan[i].GW;
    c.GH = plan[i].GH; c.GW = plan[i].GW;
    c.ntaps = plan[i].ntaps;
    c.K = c.ntaps * C;
    c.nsteps = tdg_ceil_div(c.K, bke);
    c.Kp = Kp;
    c.oh0 = plan[i].oh0; c.ow0 = plan[i].ow0;
    c.w_off_bytes = off;
    c.fd_ghw = make_fastdiv(c.GH * c.GW);
    c.fd_gw = make_fastdiv(c.GW);
    for (int t = 0; t < c.ntaps; ++t) c.tap[t] = pack_tap(plan[i].dh[t], plan[i].dw[t]);
    c.nh = plan[i].nh; c.nw = plan[i].nw; c.dh0 = plan[i].dh[0]; c.dw0 = plan[i].dw[0]; c.sh = c.sw = -1;
    c.fd_nw = make_fastdiv(c.nw);
    c.fd_nt = make_fastdiv(c.ntaps);
    off += (unsigned)((size_t)d->c * Kp * es);
  }
  const int bn = pick_bn(d->c);
  t_flops = conv_flops(d, n_images);
  return d->dtype == TDG_BF16 ? launch_fwd<bf16_t>(a, veca, bn, (hipStream_t)stream)
                              : launch_fwd<float>(a, veca, bn, (hipStream_t)stream);
}

// large bf16 filter gradients take the LDS-DMA kernel (256 x 208 tiles, one workgroup per CU)
static bool wgrad_use_dma(const TdgConvDesc* d) {
  const char* e = getenv("TDG_WDMA");                   // diagnostics: 0 disables
  if (e && atoi(e) == 0) return false;
  if (d->dtype != TDG_BF16) return false;
  const int ce = eff_channels(d->c, d->cs, 8);
  static const int min_kk = getenv("TDG_WDMA_MINKK") ? atoi(getenv("TDG_WDMA_MINKK")) : 128;    // diagnostics
  const int bn = pick_bn(d->k);
  return ce != 0 && (bn == 208 || (bn == 128 && d->k > 112)) && (long long)d->kh * d->kw * ce >= min_kk;
}

// column tile of the filter-gradient GEMM: pick_bn's, except that the LDS-DMA kernel takes 256-column tiles for N % 256 == 0
// (pix2pix / VAE widths 256, 512, 1024: a third fewer operand bytes per MAC than 256 x 128)
static int wgrad_bn(const TdgConvDesc* d) {
  const int bn = pick_bn(d->k);
  static const int wide = getenv("TDG_WG256") ? atoi(getenv("TDG_WG256")) : 1;     // diagnostics: 0 = 128-column tiles
  if (wide && bn == 128 && wgrad_use_dma(d) && d->k % 256 == 0) return 256;
  return bn;
}

static int wgrad_nsplit(const TdgConvDesc* d, int n_images, int* m_per_split) {
  const int es = tdg_dtype_size(d->dtype), vec = 16 / es;
  const int mr = d->dtype == TDG_BF16 ? WgGeom<bf16_t>::MR : WgGeom<float>::MR;
  int ce = eff_channels(d->c, d->cs, vec);
  if (!ce) ce = d->c;
  const int M = n_images * d->oh * d->ow;
  const bool dma = wgrad_use_dma(d);
  const int tiles = tdg_ceil_div((long long)d->kh * d->kw * ce, dma ? 256 : 128) * tdg_ceil_div(d->k, wgrad_bn(d));
  int want = tdg_ceil_div(768, tiles);                  // register-staged kernel: ~3 workgroups per CU
  if (dma) {
    // one workgroup per CU: the fewest splits (each costs an f32 slab written and re-read) whose last round of
    // 256 workgroups is within 10 % of the best fill any split count up to 16 (few tiles: up to 256 / tiles) reaches
    static const int force = getenv("TDG_WSPLIT") ? atoi(getenv("TDG_WSPLIT")) : 0;   // diagnostics
    double best = 0.0;
    const int sp_max = tiles >= 16 ? 16 : 256 / tiles;    // few tiles: up to one round of splits
    for (int sp = 1; sp <= sp_max; ++sp) {
      const double fill = (double)tiles * sp / (256.0 * tdg_ceil_div((long long)tiles * sp, 256));
      best = fill > best ? fill : best;
    }
    want = sp_max;
    for (int sp = 1; sp <= sp_max; ++sp) {
      const double fill = (double)tiles * sp / (256.0 * tdg_ceil_div((long long)tiles * sp, 256));
      if (fill >= 0.9 * best) { want = sp; break; }
    }
    if (force) want = force;
  }
  const int max_split = tdg_ceil_div(M, mr * 4);        // keep >= 4 steps per split
  if (want > max_split) want = max_split;
  if (want < 1) want = 1;
  if (want > 256) want = 256;
  // 208-column LDS-DMA problems: splits of whole ring cycles (3 steps) of the patch-resident kernel, which runs its
  // unrolled-by-stage loop to a multiple of 3 steps
  const int unit = dma && wgrad_bn(d) == 208 ? 3 * mr : mr;
  int per = (int)tdg_round_up(tdg_ceil_div(M, want), unit);
  *m_per_split = per;
  return tdg_ceil_div(M, per);
}

size_t tdg_conv2d_bwd_filter_workspace_bytes(const TdgConvDesc* d, int n_images) {
  if (validate_desc(d, "tdg_conv2d_bwd_filter_workspace_bytes") != TDG_OK) return 0;
  int per;
  const int ns = wgrad_nsplit(d, n_images, &per);
  return (size_t)ns * (size_t)tdg_round_up((long long)d->kh * d->kw * d->c * d->k, 4) * sizeof(float);
}

static int bwd_filter_impl(const TdgConvDesc* d, int n_images, const void* x, int n_first, const void* x2, const void* y,
                           float* dw, float beta, void* workspace, size_t workspace_bytes, void* stream);

int tdg_conv2d_bwd_filter(const TdgConvDesc* d, int n_images, const void* x, const void* y, float* dw,
                          float beta, void* workspace, size_t workspace_bytes, void* stream) {
  return bwd_filter_impl(d, n_images, x, n_images, nullptr, y, dw, beta, workspace, workspace_bytes, stream);
}

int tdg_conv2d_bwd_filter2(const TdgConvDesc* d, int n_images, const void* x, int n_first, const void* x2, const void* y,
                           float* dw, float beta, void* workspace, size_t workspace_bytes, void* stream) {
  TDG_CHECK_ARG(n_first > 0 && n_first < n_images && x2, "tdg_conv2d_bwd_filter2: n_first %d outside (0, %d) or null x2", n_first, n_images);
  // one launch when the LDS-DMA kernel applies and the switch row is even (a 2-row DMA instruction never straddles it)
  if (wgrad_use_dma(d) && ((long long)n_first * d->oh * d->ow) % 2 == 0)
    return bwd_filter_impl(d, n_images, x, n_first, x2, y, dw, beta, workspace, workspace_bytes, stream);
  int rc = bwd_filter_impl(d, n_first, x, n_first, nullptr, y, dw, beta, workspace, workspace_bytes, stream);
  if (rc) return rc;
  const char* y2 = static_cast<const char*>(y) + (size_t)n_first * d->oh * d->ow * d->ks * tdg_dtype_size(d->dtype);
  return bwd_filter_impl(d, n_images - n_first, x2, n_images - n_first, nullptr, y2, dw, 1.0f, workspace, workspace_bytes, stream);
}

static int bwd_filter_impl(const TdgConvDesc* d, int n_images, const void* x, int n_first, const void* x2, const void* y,
                           float* dw, float beta, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = validate_desc(d, "tdg_conv2d_bwd_filter");
  if (rc) return rc;
  TDG_CHECK_ARG(n_images > 0 && n_images <= d->n, "tdg_conv2d_bwd_filter: n_images %d outside (0, %d]", n_images, d->n);
  TDG_CHECK_ARG(x && y && dw && workspace, "tdg_conv2d_bwd_filter: null pointer");
  const size_t need = tdg_conv2d_bwd_filter_workspace_bytes(d, n_images);
  if (workspace_bytes < need) {
    tdg_set_error("tdg_conv2d_bwd_filter: workspace %zu < %zu bytes", workspace_bytes, need);
    return TDG_EWORKSPACE;
  }
  const int es = tdg_dtype_size(d->dtype), vec = 16 / es;
  const int ce = eff_channels(d->c, d->cs, vec);
  const bool veca = ce != 0;
  const int C = veca ? ce : d->c;
  // the dense operand (small side) is always read with 16-byte vectors
  TDG_CHECK_ARG(d->ks % vec == 0, "tdg_conv2d_bwd_filter: small-side channel stride %d not a multiple of %d", d->ks, vec);
  TDG_CHECK_ARG(((uintptr_t)y & 15) == 0 && (!veca || ((uintptr_t)x & 15) == 0), "tdg_conv2d_bwd_filter: operands must be 16-byte aligned");
  WgArgs a;
  memset(&a, 0, sizeof(a));
  a.src = x;
  a.g = y;
  a.slabs = static_cast<float*>(workspace);
  a.src_bytes = (unsigned)((long long)n_first * d->h * d->w * d->cs * es);
  a.src2 = x2;
  a.src2_bytes = (unsigned)((long long)(n_images - n_first) * d->h * d->w * d->cs * es);
  a.g_bytes = (unsigned)((long long)n_images * d->oh * d->ow * d->ks * es);
  a.M = n_images * d->oh * d->ow;
  a.m_switch = n_first * d->oh * d->ow;
  a.img_switch = n_first;
  TDG_CHECK_ARG(!x2 || ((uintptr_t)x2 & 15) == 0, "tdg_conv2d_bwd_filter2: x2 must be 16-byte aligned");
  a.GH = d->oh; a.GW = d->ow; a.SH = d->h; a.SW = d->w; a.sigma = d->stride;
  a.C = C; a.Clog = d->c; a.Cs = d->cs; a.ntaps = d->kh * d->kw;
  a.KK = a.ntaps * C;
  a.fd_c = make_fastdiv(veca ? C / vec : C);
  a.fd_ghw = make_fastdiv(d->oh * d->ow);
  a.fd_gw = make_fastdiv(d->ow);
  a.N = (int)tdg_round_up(d->k, vec) <= d->ks ? (int)tdg_round_up(d->k, vec) : d->k;
  a.Gs = d->ks;
  a.Nlog = d->k;
  a.nsplit = wgrad_nsplit(d, n_images, &a.m_per_split);
  a.slab_stride = tdg_round_up((long long)a.ntaps * d->c * d->k, 4);
  const int bn = wgrad_bn(d);
  const bool dma = wgrad_use_dma(d);
  a.ntiles_n = tdg_ceil_div(a.N, bn);
  a.ntiles_k = tdg_ceil_div(a.KK, dma ? 256 : 128);
  for (int kh = 0; kh < d->kh; ++kh)
    for (int kw = 0; kw < d->kw; ++kw) a.tap[kh * d->kw + kw] = pack_tap(kh - d->pad_t, kw - d->pad_l);
  t_flops = conv_flops(d, n_images);
  // loader mode of the gathered operand (WdLoader): 1 = a 64-row step of whole images, 2 = a rectangle of one image
  // (whole grid rows, or a 64-column piece of one), 0 = anything else.  TDG_WDMA_SLOWA (diagnostics) forces 0.
  const int ghw = d->oh * d->ow;
  int mode = 0;
  if (WD_MR % ghw == 0) mode = 1;
  else if (ghw % WD_MR == 0 && (d->ow % WD_MR == 0 || WD_MR % d->ow == 0) && a.m_switch % WD_MR == 0 &&
           d->pad_t <= WD_HALO && d->pad_l <= WD_HALO) mode = 2;
  if (getenv("TDG_WDMA_SLOWA")) mode = 0;
  auto go = [&](auto bn_c) -> int {
    constexpr int B = decltype(bn_c)::value;
    return mode == 1 ? launch_wgrad_dma<B, 1>(a, (hipStream_t)stream)
                     : (mode == 2 ? launch_wgrad_dma<B, 2>(a, (hipStream_t)stream) : launch_wgrad_dma<B, 0>(a, (hipStream_t)stream));
  };
  // whole-image steps of a 208-column problem: the patch-resident kernel (tdg_wgrad_patch.hip) where its plan applies.
  // TDG_WPATCH=0 (variant tests) keeps the slab kernel.
  bool patched = false;
  if (dma && bn == 208 && mode == 1 && !(getenv("TDG_WPATCH") && atoi(getenv("TDG_WPATCH")) == 0)) {
    WpPlan wp;
    if (tdg_wgrad_patch_plan(a, &wp)) {
      rc = tdg_wgrad_patch_launch(a, wp, t_flops, (hipStream_t)stream);
      if (rc) return rc;
      patched = true;
    }
  }
  if (!patched)
  rc = dma ? (bn == 208 ? go(std::integral_constant<int, 208>{}) : bn == 256 ? go(std::integral_constant<int, 256>{}) : go(std::integral_constant<int, 128>{}))
           : d->dtype == TDG_BF16 ? launch_wgrad<bf16_t>(a, veca, bn, (hipStream_t)stream)
                                  : launch_wgrad<float>(a, veca, bn, (hipStream_t)stream);
  if (rc) return rc;
  const size_t n = (size_t)a.ntaps * d->c * d->k;
  tdg_timing_start("slab_reduce", 0.0, (hipStream_t)stream);
  if (a.nsplit <= 8 && ((uintptr_t)dw & 15) == 0) {
    hipLaunchKernelGGL(slab_reduce_few_kernel, dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a.slabs,
                       dw, n, a.nsplit, (size_t)a.slab_stride, beta);
  } else {
    const int blocks = (int)(((n + 3) / 4 + 15) / 16);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a.slabs, dw, n, a.nsplit,
                       (size_t)a.slab_stride, beta);
  }
  tdg_timing_stop((hipStream_t)stream);
  TDG_HIP_LAUNCH_CHECK("slab_reduce");
  return TDG_OK;
}

}  // extern "C"
