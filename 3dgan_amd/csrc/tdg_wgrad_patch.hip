// lib3dgan_hip.so -- patch-resident filter-gradient GEMM for gfx950 (MI355X), bf16.
//
//   dW[(tap, c)][n] = sum_m x[pixel(m) + tap][c] * dy[m][n]          (conv2d_backprop_filter: autodiff of
//                                                                    /root/reference/ops/layers.py:101,142)
//
// igemm_wgrad_dma_kernel streams 64 im2col rows x 256 filter rows per step (32 KB) beside 64 x 208 columns of dy; with
// 59 - 64 KB per 6.8 MFLOP step the CU's L2 -> LDS intake (~30 - 36 B/clk) takes as long as the step's MFMAs and the
// waves that issue the LDS-DMA pieces stall in front of a full queue.  But the im2col rows of a step are re-reads: a 5x5
// stride-2 filter touches every source pixel for 6.25 taps.  This kernel stages the SOURCE PIXELS of a step once:
//   * K is ordered in units of (8-channel slice, tap); a 256-row tile of dW = 32 consecutive units spans <= nsl slices;
//   * a step = 64 rows of dy = 64 / (GH*GW) whole images; its patch [image][row][column][nsl chunks of 16 B] (12 KB for
//     the GAN's 16x16 -> 8x8 and 8x8 -> 4x4 layers) arrives by LDS-DMA with per-lane constant offsets and a descriptor
//     that moves by whole images per step (scalar ALU only);
//   * the transposing fragment read `ds_read_b64_tr_b16` takes a per-lane ADDRESS for each of its 4 rows x 4 column
//     quads, so it gathers the im2col rows straight out of the patch: lane (row q, quad p) reads 8 bytes of pixel
//     (row q's anchor + tap(unit p>>1)), channels 4(p&1).. of the unit's slice, or 8 zero bytes at the head of the ring stage when the tap
//     falls outside the image.  All addresses are loop invariants (one VGPR per fragment half); the ring stage is the
//     instruction's immediate offset, so the K loop contains NO vector ALU instruction besides the MFMAs;
//   * dy rows are 416-byte LDS rows (208 columns, no padding chunks: 26 pieces per step instead of 32), conflict-free
//     for the transposing reads because a 32-lane half reads 8 consecutive rows (416 B = 104 dwords = 40 mod 64);
//   * 3-stage ring (3 x (16 + 26) KB), the pieces of step s+3 are issued behind the barrier of step s, a wave waits for
//     its own pieces of step s+1 with a counted vmcnt: two steps of slack for an L2 miss.
// Intake per step: 12 + 26 KB instead of 32 + 32.  Tile 256 (filter rows) x 208 (columns); NW = 8 waves as 4 x 2 (two per
// SIMD, 256 registers) or NW = 4 waves as 2 x 2 (one per SIMD, 128 x 104 accumulators in a 512-register wave).
#include <stdlib.h>

#include <type_traits>

#include "tdg_wgrad_patch.h"

#define WP_OOB 0xFFFFFF00u
#define WP_MR 64                                   // rows of dy per step
#define WP_GCH 26                                  // 16-byte chunks per dy row in LDS (208 columns)
#define WP_GROWB (WP_GCH * 16)
#define WP_GPIECES (WP_MR * WP_GCH / 64)           // 26
#define WP_GSTAGE (WP_GPIECES * 1024)
#define WP_ASTAGE 16384                            // a patch stage: 1 KiB of zeros + <= 15 pieces
#define WP_NST 3
#define WP_ZERO 1024                               // the first KiB of every patch stage stays zero: the ring stage is an immediate
                                                   // offset of the fragment reads, so a tap outside the image reads byte 0 of ITS stage
#define WP_AOFF 0
#define WP_GOFF (WP_AOFF + WP_NST * WP_ASTAGE)
#define WP_LDS (WP_GOFF + WP_NST * WP_GSTAGE)      // 130048 (+ the tap table)
#define WP_MAXSL 5

namespace {
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) bf16x4* lds_b4_t;
typedef __attribute__((address_space(3))) char* lds_c_t;

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(IntC<I>{});
    sfor<I + 1, N>(f);
  }
}

// LDS-DMA from inline asm (the builtin form makes hipcc drain vmcnt in front of every transposing read)
__device__ __forceinline__ void wp_dma(const i32x4& rsrc, unsigned voff, unsigned lds_byte) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_byte) : "memory");
}
// a buffer window [p, p + 16 rem) that moves by a constant per step (scalar registers only; operand sizes and steps are
// multiples of 16 bytes and tensors are < 4 GB, so the remainder counts 16-byte chunks in an int: a scalar compare)
struct WpCursor {
  unsigned long long p;
  int rem;
  __device__ __forceinline__ void advance(int step16) { p += (unsigned long long)((unsigned)step16 << 4); rem -= step16; }
  __device__ __forceinline__ i32x4 desc() const {
    const unsigned r = rem > 0 ? (unsigned)rem << 4 : 0u;
    return i32x4{(int)(unsigned)p, (int)((unsigned)(p >> 32) & 0xffffu), (int)r, 0x00020000};
  }
};
struct WpDescs { i32x4 g, a; };

__device__ __forceinline__ bf16x8 wp_frag(unsigned lo_addr, unsigned hi_addr) {
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_t)((lds_c_t) nullptr + lo_addr));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4_t)((lds_c_t) nullptr + hi_addr));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ void wp_mfma_agpr(f32x4& c, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
}  // namespace

template <int NW, int NSLOT>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 2 : 1) igemm_wgrad_patch_kernel(const WgArgs args, const WpPlan pl) {
  constexpr int WK = NW / 2;                 // waves along the filter rows
  constexpr int RT = 16 / WK;                // 16-row tiles per wave
  constexpr int TN = 7, TN1 = 6;             // column tiles of the two wave columns (13 x 16 = 208)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
  int* sTap = reinterpret_cast<int*>(smem + WP_LDS);

  const int tid = threadIdx.x;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_k = bid / args.ntiles_n;
  const int tile_n = bid - tile_k * args.ntiles_n;
  const int n0 = tile_n * 208;
  const int split = blockIdx.z;
  const int m_begin = split * args.m_per_split;
  const int m_end = min(args.M, m_begin + args.m_per_split);
  const int nsteps = (m_end - m_begin + WP_MR - 1) / WP_MR;
  const int SH = args.SH, SW = args.SW, Cs = args.Cs, sigma = args.sigma, Gs = args.Gs, GHW = args.GH * args.GW, GW = args.GW;
  const int ntaps = args.ntaps;

  for (int i = tid; i < WP_NST * (WP_ZERO / 16); i += 64 * NW)
    *reinterpret_cast<f32x4*>(smem + (i / (WP_ZERO / 16)) * WP_ASTAGE + (i % (WP_ZERO / 16)) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid < IG_MAX_TAPS) sTap[tid] = args.tap[tid];
  __syncthreads();

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wk = wave % WK, wn = wave / WK;
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int u0 = tile_k * 32;
  const int s0 = (int)fd_div((unsigned)u0, pl.fd_nt);        // first slice of this tile's patch

  // ---- loop-invariant fragment addresses --------------------------------------------------------------------------
  // MFMA k index (lane group g, element e = 4h + q') of slice ks  <->  row 32 ks + 16 h + 8 (g >> 1) + 4 (g & 1) + q' of the
  // step: both operands use this map, and a 32-lane half of one read covers 8 consecutive rows.
  unsigned addrA[RT][2][2];
  {
    int ih0[2][2], iw0[2][2], prow0[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const unsigned ml = 32 * ks + 16 * h + 8 * (g >> 1) + 4 * (g & 1) + q;
        const unsigned il = fd_div(ml, args.fd_ghw);
        const unsigned rem = ml - il * (unsigned)GHW;
        const unsigned oh = fd_div(rem, args.fd_gw);
        const unsigned ow = rem - oh * (unsigned)GW;
        ih0[ks][h] = (int)oh * sigma;
        iw0[ks][h] = (int)ow * sigma;
        prow0[ks][h] = (int)il * SH;
      }
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const int u = u0 + 2 * (wk * RT + i) + (p >> 1);
      const bool uok = u < pl.nunits;
      const unsigned uu = uok ? (unsigned)u : 0u;
      const unsigned sl = fd_div(uu, pl.fd_nt);
      const int tap = (int)(uu - sl * (unsigned)ntaps);
      const int pk = sTap[tap];
      const int dh = (pk << 24) >> 24, dw = (pk << 16) >> 24;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ih = ih0[ks][h] + dh, iw = iw0[ks][h] + dw;
          const bool ok = uok && (unsigned)ih < (unsigned)SH && (unsigned)iw < (unsigned)SW;
          const int chunk = (prow0[ks][h] + ih) * pl.rp + iw * pl.nsl + ((int)sl - s0);
          // (a tap outside the image reads zeros from the stage's zero block, at the bank its pixel would have had: the
          // layout search below then sees the same conflict pattern as for an interior tile)
          const unsigned in_patch = (unsigned)(chunk << 4) + (unsigned)((p & 1) << 3);
          addrA[i][ks][h] = lds0 + (ok ? (unsigned)(WP_AOFF + WP_ZERO) + in_patch : (unsigned)WP_AOFF + (in_patch & 0xf8u));
        }
    }
  }
  unsigned gaddr[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
    gaddr[ks] = lds0 + (unsigned)(WP_GOFF + (32 * ks + 8 * (g >> 1) + 4 * (g & 1) + q) * WP_GROWB + p * 8 + wn * TN * 32);

  // ---- the loader: slot k of wave w is piece w + k NW of [26 dy pieces | npa patch pieces | dummies] ---------------
  unsigned vo[NSLOT];
  sfor<0, NSLOT>([&](auto k_c) {
    constexpr int k = decltype(k_c)::value;
    const int P = wave + k * NW;
    if (P < WP_GPIECES) {
      const unsigned idx = (unsigned)(64 * P + lane);
      const unsigned row = idx / WP_GCH, ch = idx - row * WP_GCH;
      const int n = n0 + (int)ch * 8;
      vo[k] = n < args.N ? (unsigned)((int)row * Gs + n) * 2u : WP_OOB;
    } else {
      const int PA = P - WP_GPIECES;
      const unsigned idx = (unsigned)(64 * PA + lane);
      const unsigned prow = fd_div(idx, pl.fd_rp);
      const unsigned rc = idx - prow * (unsigned)pl.rp;
      const unsigned iw = fd_div(rc, pl.fd_nsl);
      const unsigned sl = rc - iw * (unsigned)pl.nsl;
      const bool ok = PA < pl.npa && (int)prow < pl.imgs * SH && (int)iw < SW && s0 + (int)sl < pl.nslices;
      vo[k] = ok ? (unsigned)(((int)prow * SW + (int)iw) * Cs + (s0 + (int)sl) * 8) * 2u : WP_OOB;
    }
  });
  const int wave_kb = (int)lds0 + wave * 1024;      // scalar

  struct Cursors { WpCursor a1, a2, g; };
  Cursors cur;
  int stepA, stepG;                          // 16-byte chunks per step
  {
    const long long img16 = (long long)SH * SW * Cs / 8;
    const int m_sw = m_end < args.m_switch ? m_end : args.m_switch;
    const void* s2 = args.src2 ? args.src2 : args.src;
    cur.a1.p = (unsigned long long)args.src + (unsigned long long)(m_begin / GHW) * (unsigned long long)(img16 * 16);
    cur.a1.rem = (int)((long long)(m_sw / GHW - m_begin / GHW) * img16);
    cur.a2.p = (unsigned long long)s2 + (unsigned long long)((long long)(m_begin / GHW - args.img_switch) * img16 * 16);
    cur.a2.rem = (int)((long long)(m_end / GHW - m_begin / GHW) * img16);
    cur.g.p = (unsigned long long)args.g + (unsigned long long)m_begin * (unsigned long long)(Gs * 2);
    cur.g.rem = (int)((long long)(m_end - m_begin) * Gs / 8);
    stepA = (int)(pl.imgs * img16);
    stepG = WP_MR * Gs / 8;
  }
  auto descs = [&](int mstep) -> WpDescs {
    WpDescs d;
    d.g = cur.g.desc();
    const i32x4 d1 = cur.a1.desc(), d2 = cur.a2.desc();
    d.a = mstep >= args.m_switch ? d2 : d1;            // m_switch is a multiple of the step: wave-uniform, scalar select
    return d;
  };
  auto advance = [&]() {
    cur.g.advance(stepG);
    cur.a1.advance(stepA);
    cur.a2.advance(stepA);
  };
  // (dummy slots -- beyond the patch's pieces -- write their zeros onto the zero block of stage 0)
  auto piece = [&](auto k_c, auto st_c, const WpDescs& d) {
    constexpr int k = decltype(k_c)::value, st = decltype(st_c)::value;
    constexpr int GD = WP_GOFF + st * WP_GSTAGE + k * NW * 1024, AD = WP_AOFF + st * WP_ASTAGE + WP_ZERO + (k * NW - WP_GPIECES) * 1024;
    if constexpr (k * NW + NW - 1 < WP_GPIECES) {
      wp_dma(d.g, vo[k], (unsigned)(wave_kb + GD));
    } else if constexpr (k * NW >= WP_GPIECES) {
      const bool real = wave + k * NW - WP_GPIECES < pl.npa;
      wp_dma(d.a, vo[k], (unsigned)(real ? wave_kb + AD : (int)lds0));
    } else {
      const bool isg = wave + k * NW < WP_GPIECES;
      const bool real = wave + k * NW - WP_GPIECES < pl.npa;
      wp_dma(isg ? d.g : d.a, vo[k], (unsigned)(isg ? wave_kb + GD : (real ? wave_kb + AD : (int)lds0)));
    }
  };

  f32x4 acc[RT][TN];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- main loop ------------------------------------------------------------------------------------------------
  // Step s lives in ring stage s % 3.  Fragments are double-buffered in registers by 32-row slice; the step's barrier is
  // SKEWED: it sits TS column tiles into slice 1.  Behind barrier B_s every wave has finished reading stage s % 3 (all of
  // slice 1 was requested during slice 0) and every wave's pieces of step s+1 have landed, so the tail of step s requests
  // the slice-0 fragments of step s+1 and issues the first pieces of step s+3 into stage s % 3; the rest of those pieces
  // ride on the slice-0 tiles of step s+1.  Rows past m_end are out-of-range sources (zero fill): no peeled last step.
  auto run = [&](auto tnw_c) {
    constexpr int TNW = decltype(tnw_c)::value;
    constexpr int TS = TNW >= 6 ? 3 : TNW / 2;
    constexpr int NTAIL = TNW - TS;
    constexpr int NPB = NSLOT / 2 < NTAIL ? NSLOT / 2 : NTAIL;      // pieces on the tail tiles
    constexpr int NPA = NSLOT - NPB;                                // pieces on the slice-0 tiles
    constexpr int PPT = (NPA + TNW - 1) / TNW;
    bf16x8 FA[2][RT], FG[2][TNW];
    auto mma_tile = [&](auto ks_c, auto t_c) {
      constexpr int ks = decltype(ks_c)::value, t = decltype(t_c)::value;
      if constexpr (NW == 4) {
        // one wave per SIMD, 512 registers: the 56 accumulators are pinned to AGPRs by the asm constraint (given the builtin,
        // hipcc renames them between the unrolled stage bodies and fills the loop with v_accvgpr moves: 432 per 288 MFMAs)
#pragma unroll
        for (int i = 0; i < RT; ++i)
          wp_mfma_agpr(acc[i][t], FG[ks][t], FA[ks][i]);
      } else {
#pragma unroll
        for (int i = 0; i < RT; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FG[ks][t], FA[ks][i], acc[i][t], 0, 0, 0);
      }
    };
    auto readA = [&](auto st_c, auto ks_c, auto i_c) {
      constexpr int st = decltype(st_c)::value, ks = decltype(ks_c)::value, i = decltype(i_c)::value;
      FA[ks][i] = wp_frag(addrA[i][ks][0] + st * WP_ASTAGE, addrA[i][ks][1] + st * WP_ASTAGE);
    };
    auto readG = [&](auto st_c, auto ks_c, auto t_c) {
      constexpr int st = decltype(st_c)::value, ks = decltype(ks_c)::value, t = decltype(t_c)::value;
      FG[ks][t] = wp_frag(gaddr[ks] + t * 32 + st * WP_GSTAGE, gaddr[ks] + 16 * WP_GROWB + t * 32 + st * WP_GSTAGE);
    };

    // prologue: steps 0 and 1 whole, the tail-tile pieces of step 2; then the slice-0 fragments of step 0
    WpDescs d2;
    {
      const WpDescs da = descs(m_begin);
      sfor<0, NSLOT>([&](auto k_c) { piece(k_c, IntC<0>{}, da); });
      advance();
      const WpDescs db = descs(m_begin + WP_MR);
      sfor<0, NSLOT>([&](auto k_c) { piece(k_c, IntC<1>{}, db); });
      advance();
      d2 = descs(m_begin + 2 * WP_MR);
      sfor<0, NPB>([&](auto k_c) { piece(k_c, IntC<2>{}, d2); });
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSLOT + NPB) : "memory");
    __syncthreads();
    sfor<0, RT>([&](auto i_c) { readA(IntC<0>{}, IntC<0>{}, i_c); });
    sfor<0, TNW>([&](auto t_c) { readG(IntC<0>{}, IntC<0>{}, t_c); });

    int mstep3 = m_begin + 3 * WP_MR;                   // first row of step s + 3
    auto body = [&](auto st_c) {
      constexpr int ST = decltype(st_c)::value, ST1 = (ST + 1) % WP_NST, ST2 = (ST + 2) % WP_NST;
      // ---- part A: slice 0 (+ the reads of slice 1, + the remaining pieces of step s+2), then slice-1 tiles [0, TS)
      sfor<0, TNW>([&](auto t_c) {
        constexpr int t = decltype(t_c)::value;
        constexpr int a_lo = (t * RT) / TNW, a_hi = ((t + 1) * RT) / TNW;
        readG(st_c, IntC<1>{}, t_c);
        sfor<a_lo, a_hi>([&](auto i_c) { readA(st_c, IntC<1>{}, i_c); });
        mma_tile(IntC<0>{}, t_c);
        constexpr int p_lo = NPB + t * PPT < NSLOT ? NPB + t * PPT : NSLOT;
        constexpr int p_hi = NPB + (t + 1) * PPT < NSLOT ? NPB + (t + 1) * PPT : NSLOT;
        sfor<p_lo, p_hi>([&](auto k_c) { piece(k_c, IntC<ST2>{}, d2); });
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * (1 + a_hi - a_lo), 0);
        __builtin_amdgcn_sched_group_barrier(0x008, RT, 0);
        if constexpr (p_hi > p_lo) __builtin_amdgcn_sched_group_barrier(0x020, p_hi - p_lo, 0);
      });
      sfor<0, TS>([&](auto t_c) {
        mma_tile(IntC<1>{}, t_c);
        __builtin_amdgcn_sched_group_barrier(0x008, RT, 0);
      });
      // ---- the step's barrier
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSLOT) : "memory");
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      advance();
      const WpDescs d3 = descs(mstep3);
      mstep3 += WP_MR;
      // ---- part B: the rest of slice 1; underneath, the slice-0 fragments of step s+1 and the first pieces of step s+3
      sfor<0, NTAIL>([&](auto u_c) {
        constexpr int u = decltype(u_c)::value, t = TS + u;
        constexpr int g_lo = u == 0 ? 0 : 1 + ((TNW - 1) * (u - 1)) / (NTAIL - 1 > 0 ? NTAIL - 1 : 1);
        constexpr int g_hi = u == 0 ? 1 : (u == NTAIL - 1 ? TNW : 1 + ((TNW - 1) * u) / (NTAIL - 1 > 0 ? NTAIL - 1 : 1));
        if constexpr (u == 0) sfor<0, RT>([&](auto i_c) { readA(IntC<ST1>{}, IntC<0>{}, i_c); });
        sfor<g_lo, g_hi>([&](auto g_c) { readG(IntC<ST1>{}, IntC<0>{}, g_c); });
        mma_tile(IntC<1>{}, IntC<t>{});
        if constexpr (u < NPB) piece(u_c, st_c, d3);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * ((u == 0 ? RT : 0) + (g_hi - g_lo)), 0);
        __builtin_amdgcn_sched_group_barrier(0x008, RT, 0);
        if constexpr (u < NPB) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      });
      d2 = d3;
    };
    // (the loop is unrolled by the ring's three stages with ONE exit: a trip count that is not a multiple of 3 runs up to two
    // extra steps on zero-fill pieces -- exits between the stage bodies made hipcc rename the accumulators per body, which
    // the 512-register form cannot afford; the host picks splits whose step count is a multiple of 3)
    for (int s = 0; s < nsteps; s += WP_NST) {
      body(IntC<0>{});
      body(IntC<1>{});
      body(IntC<2>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the trailing (out-of-range, zero-fill) pieces
    if constexpr (NW == 4) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the asm MFMAs' results (no compiler-known hazard) before the epilogue reads them
  };
  if (wn == 0) run(IntC<TN>{});
  else run(IntC<TN1>{});

  // ---- epilogue: lane owns filter row (unit, channel r16 & 7) x 4 consecutive n ----------------------------------
  const int r16 = lane & 15, qq = lane >> 4;
  const int tnw = wn == 0 ? TN : TN1;
  float* slab = args.slabs + (size_t)split * (size_t)args.slab_stride;
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int u = u0 + 2 * (wk * RT + i) + (r16 >> 3);
    if (u >= pl.nunits) continue;
    const int sl = (int)fd_div((unsigned)u, pl.fd_nt);
    const int tap = u - sl * ntaps;
    const int c = sl * 8 + (r16 & 7);
    if (c >= args.Clog) continue;
    float* rowp = slab + ((size_t)tap * args.Clog + c) * (size_t)args.Nlog;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 16 + qq * 4;
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
}

// ------------------------------------------------------------------------------------------------------------ host
// extra LDS cycles of the A-fragment reads of tile 0 for a patch row pitch (the bank of an 8-byte access is its 8-byte
// granule mod 32; lanes of a 32-lane half reading different addresses in one granule class serialise)
static int wp_conflicts(const WgArgs& a, int nsl, int rp) {
  const int ghw = a.GH * a.GW;
  int extra = 0;
  for (int i = 0; i < 16; ++i)
    for (int ks = 0; ks < 2; ++ks)
      for (int h = 0; h < 2; ++h)
        for (int half = 0; half < 2; ++half) {
          unsigned addr[32];
          int n = 0;
          for (int gl = 0; gl < 2; ++gl)
            for (int q = 0; q < 4; ++q)
              for (int p = 0; p < 4; ++p) {
                const int g = 2 * half + gl;
                const int ml = 32 * ks + 16 * h + 8 * (g >> 1) + 4 * (g & 1) + q;
                const int il = ml / ghw, rem = ml % ghw, oh = rem / a.GW, ow = rem % a.GW;
                const int u = 2 * i + (p >> 1);
                const int sl = u / a.ntaps, tap = u % a.ntaps;
                const int dh = (signed char)(a.tap[tap] & 0xff), dw = (signed char)((a.tap[tap] >> 8) & 0xff);
                const int ih = oh * a.sigma + dh, iw = ow * a.sigma + dw;
                const bool ok = u < (a.C / 8) * a.ntaps && ih >= 0 && ih < a.SH && iw >= 0 && iw < a.SW;
                const unsigned in_patch = (unsigned)((((il * a.SH + ih) * rp + iw * nsl + sl) << 4)) + (unsigned)((p & 1) << 3);
                addr[n++] = ok ? (unsigned)(WP_AOFF + WP_ZERO) + in_patch : (unsigned)WP_AOFF + (in_patch & 0xf8u);
              }
          int worst = 1;
          for (int b = 0; b < 32; ++b) {
            unsigned seen[32];
            int ns = 0;
            for (int l = 0; l < 32; ++l) {
              if ((int)((addr[l] >> 3) & 31) != b) continue;
              bool dup = false;
              for (int k = 0; k < ns; ++k) dup |= seen[k] == addr[l];
              if (!dup) seen[ns++] = addr[l];
            }
            worst = ns > worst ? ns : worst;
          }
          extra += worst - 1;
        }
  return extra;
}

// the layout search costs ~1 ms of host time: plans are remembered per geometry (a handful per model)
struct WpKey {
  int GH, GW, SH, SW, sigma, C, ntaps, ntiles_k;
  short tap[IG_MAX_TAPS];
};
static bool wp_plan_search(const WgArgs& a, WpPlan* p);
bool tdg_wgrad_patch_plan(const WgArgs& a, WpPlan* p) {
  memset(p, 0, sizeof(*p));
  // per-launch conditions first (the remembered part depends on the geometry only)
  if (a.m_switch < a.M && a.m_switch % WP_MR != 0) return false;        // a step reads one source tensor
  if (a.m_per_split % WP_MR != 0) return false;
  if (a.Gs % 8 != 0 || a.N % 8 != 0 || a.Cs % 8 != 0) return false;
  if (a.ntiles_n != tdg_ceil_div(a.N, 208)) return false;
  WpKey key;
  memset(&key, 0, sizeof(key));
  key.GH = a.GH; key.GW = a.GW; key.SH = a.SH; key.SW = a.SW; key.sigma = a.sigma; key.C = a.C; key.ntaps = a.ntaps; key.ntiles_k = a.ntiles_k;
  memcpy(key.tap, a.tap, sizeof(key.tap));
  struct Entry { WpKey key; WpPlan plan; bool ok; };
  static thread_local Entry cache[32];
  static thread_local int n_cached = 0, next = 0;
  for (int i = 0; i < n_cached; ++i)
    if (!memcmp(&cache[i].key, &key, sizeof(key))) {
      *p = cache[i].plan;
      return cache[i].ok;
    }
  const bool ok = wp_plan_search(a, p);
  Entry& e = cache[next];
  next = (next + 1) % 32;
  if (n_cached < 32) ++n_cached;
  e.key = key; e.plan = *p; e.ok = ok;
  return ok;
}

static bool wp_plan_search(const WgArgs& a, WpPlan* p) {
  memset(p, 0, sizeof(*p));
  const int ghw = a.GH * a.GW;
  if (ghw <= 0 || WP_MR % ghw != 0) return false;                       // a step = whole images
  if (a.C <= 0 || a.C % 8 != 0 || a.Cs % 8 != 0) return false;          // 16-byte channel slices
  if (a.ntaps < 1 || a.ntaps > IG_MAX_TAPS) return false;
  p->nslices = a.C / 8;
  p->nunits = p->nslices * a.ntaps;
  if (a.ntiles_k != tdg_ceil_div(p->nunits, 32)) return false;
  int nsl = 1;
  for (int t = 0; t < a.ntiles_k; ++t) {
    const int last = 32 * t + 31 < p->nunits ? 32 * t + 31 : p->nunits - 1;
    const int span = last / a.ntaps - (32 * t) / a.ntaps + 1;
    nsl = span > nsl ? span : nsl;
  }
  if (nsl > WP_MAXSL) return false;
  p->nsl = nsl;
  p->imgs = WP_MR / ghw;
  const int rows = p->imgs * a.SH;
  int best_rp = 0, best = 1 << 30;
  for (int rp = a.SW * nsl; rp <= a.SW * nsl + 15; ++rp) {
    if ((long long)rows * rp * 16 > WP_ASTAGE - WP_ZERO) break;
    const int c = wp_conflicts(a, nsl, rp);
    if (c < best) { best = c; best_rp = rp; }
    if (c == 0) break;
  }
  if (!best_rp) return false;                                            // the patch does not fit a ring stage
  p->rp = best_rp;
  p->conflicts = best;
  p->npa = tdg_ceil_div((long long)rows * best_rp, 64);
  p->fd_nt = make_fastdiv((uint32_t)a.ntaps);
  p->fd_rp = make_fastdiv((uint32_t)best_rp);
  p->fd_nsl = make_fastdiv((uint32_t)nsl);
  return true;
}

template <int NW, int NSLOT>
static int wp_launch(WgArgs& a, WpPlan& p, double flops, hipStream_t s) {
  const size_t lds = WP_LDS + IG_MAX_TAPS * sizeof(int);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_wgrad_patch_kernel<NW, NSLOT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  static char name[64] = "";
  if (!name[0]) snprintf(name, sizeof(name), "igemm_wgrad_patch_kernel<bf16,256,208,%d,%d>", NW, NSLOT);
  p.nslot = NSLOT;
  dim3 grid(a.ntiles_k * a.ntiles_n, 1, a.nsplit), block(64 * NW);
  tdg_note_kernel(name);
  tdg_timing_start(name, flops, s);
  hipLaunchKernelGGL((igemm_wgrad_patch_kernel<NW, NSLOT>), grid, block, lds, s, a, p);
  tdg_timing_stop(s);
  TDG_HIP_LAUNCH_CHECK("igemm_wgrad_patch");
  return TDG_OK;
}

int tdg_wgrad_patch_launch(WgArgs& a, WpPlan& p, double flops, hipStream_t s) {
  const char* e = getenv("TDG_WPATCH_NW");               // variant tests: 4 = one wave per SIMD (512-register waves)
  const int nw = e ? atoi(e) : 8;
  const int pieces = WP_GPIECES + p.npa;
  if (nw == 4) {
    const int ns = tdg_ceil_div(pieces, 4);
    if (ns <= 10) return wp_launch<4, 10>(a, p, flops, s);
    return wp_launch<4, 11>(a, p, flops, s);
  }
  const int ns = tdg_ceil_div(pieces, 8);
  if (ns <= 5) return wp_launch<8, 5>(a, p, flops, s);
  return wp_launch<8, 6>(a, p, flops, s);
}
