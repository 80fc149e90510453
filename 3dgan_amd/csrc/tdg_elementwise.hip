// HBM-bound kernels of the 3dgan hot path (gfx950): batch norm, bias / activation derivatives,
// gradient-penalty interpolation + norm, loss reductions, fused optimizer steps, Philox RNG.
// All reductions are two-stage with a fixed summation order (deterministic, no atomics);
// cross-lane sums use 64-wide wavefront shuffles.
#include <stdarg.h>

#include <atomic>

#include "tdg_common.h"
#include <type_traits>

// ---------------------------------------------------------------------------- error state
static thread_local char g_err[512] = "";
void tdg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* tdg_last_error(void) { return g_err; }
extern "C" int tdg_version(void) { return 100; }
static thread_local const char* g_last_kernel = "";
void tdg_note_kernel(const char* name) { g_last_kernel = name; }
extern "C" const char* tdg_last_kernel(void) { return g_last_kernel; }

// ---------------------------------------------------------------------------- per-launch timing (diagnostics)
// HIP events around every conv GEMM kernel launch, on the stream the kernel is launched on.  Off unless
// tdg_timing_begin() was called; not usable while a stream capture is in progress.
#include <vector>
struct TimingSlot { const char* name; double flops; hipEvent_t e0, e1; };
static std::vector<TimingSlot> g_slots;
static int g_timing_on = 0, g_timing_n = 0;
bool tdg_timing_enabled() { return g_timing_on != 0; }
void tdg_timing_start(const char* name, double flops, hipStream_t s) {
  if (!g_timing_on) return;
  if (g_timing_n == (int)g_slots.size()) {
    TimingSlot t;
    t.name = ""; t.flops = 0;
    (void)hipEventCreate(&t.e0);
    (void)hipEventCreate(&t.e1);
    g_slots.push_back(t);
  }
  TimingSlot& t = g_slots[g_timing_n];
  t.name = name;
  t.flops = flops;
  (void)hipEventRecord(t.e0, s);
}
void tdg_timing_stop(hipStream_t s) {
  if (!g_timing_on) return;
  (void)hipEventRecord(g_slots[g_timing_n].e1, s);
  ++g_timing_n;
}
extern "C" int tdg_timing_begin(void) {
  g_timing_on = 1;
  g_timing_n = 0;
  return TDG_OK;
}
extern "C" int tdg_timing_end(TdgLaunchRecord* out, int capacity, int* count) {
  g_timing_on = 0;
  TDG_CHECK_ARG(count != nullptr, "tdg_timing_end: null count");
  int n = 0;
  for (int i = 0; i < g_timing_n; ++i) {
    if (hipEventSynchronize(g_slots[i].e1) != hipSuccess) { tdg_set_error("tdg_timing_end: event sync failed"); return TDG_EHIP; }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, g_slots[i].e0, g_slots[i].e1);
    if (out && n < capacity) {
      strncpy(out[n].kernel, g_slots[i].name, sizeof(out[n].kernel) - 1);
      out[n].kernel[sizeof(out[n].kernel) - 1] = 0;
      out[n].ms = ms;
      out[n].flops = g_slots[i].flops;
    }
    ++n;
  }
  *count = n;
  g_timing_n = 0;
  return TDG_OK;
}

#define DISPATCH_T(dtype, ...)                  \
  if ((dtype) == TDG_BF16) {                    \
    using T = bf16_t;                           \
    __VA_ARGS__                                 \
  } else if ((dtype) == TDG_F32) {              \
    using T = float;                            \
    __VA_ARGS__                                 \
  } else {                                      \
    tdg_set_error("bad dtype %d", (int)(dtype));\
    return TDG_EINVAL;                          \
  }

static inline int ew_blocks(size_t n, int per_block = 1024) {
  size_t b = (n + per_block - 1) / per_block;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// block-wide sum of 256 threads; result valid in thread 0
__device__ __forceinline__ float block_sum256(float v, float* sh /*>=4*/) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) r = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return r;
}

// ---- last-block tickets ------------------------------------------------------------------------------------------
// One launch instead of two for "every block contributes, one block finishes": each block takes a ticket from a
// device-global counter when it is done; the block that draws the last ticket does the finishing work and resets the
// counter (self-cleaning, so hipGraph replays need no memset).  Kernels of one stream never overlap, and every kernel
// family has its own slot.  `last_block_ticket`: no data is handed over (the finisher only needs to know that every
// block has READ something: the Adam step count, the Philox draw counter).  `last_block_arrives`: the other blocks'
// global stores are handed over -- producer: drain, barrier, agent-scope release, ticket; finisher: agent-scope acquire,
// drain, barrier, then plain loads (MI355X_MICROARCH.md, inter-workgroup visibility).
#define TICKET_SUMSQ 1024
#define TICKET_ADAM 1025
#define TICKET_RNG 1026
#define TICKET_ADAM_SUB 1088
#define TICKET_RNG_SUB 1152
__device__ unsigned g_ticket[1216];

// The slots are per PROCESS, one per kernel family, so two launches of one family must never overlap on the device.  One
// stream guarantees that; the library therefore refuses a ticketed launch on a second stream (a stream under hipGraph
// capture is exempt: the captured launches replay in their captured order on whatever stream launches the graph, which
// the same rule covers).  Entry points concerned: tdg_sumsq, tdg_adam_step_dev, tdg_random_normal_dev,
// tdg_random_uniform_f32_dev (include/tdg.h says so at each).
static int ticket_stream_check(const char* who, void* stream) {
  // the slots (g_ticket) are per DEVICE; the remembered stream is kept per device ordinal and claimed with one atomic
  // compare-exchange, so two host threads cannot both believe they were first
  static std::atomic<void*> first[16];
  static std::atomic<bool> have[16];
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)stream, &st) != hipSuccess) {
    (void)hipGetLastError();                              // (do not leave a sticky error for the next launch check to report)
    st = hipStreamCaptureStatusNone;
  }
  if (st != hipStreamCaptureStatusNone) return TDG_OK;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  bool expected = false;
  if (have[dev].compare_exchange_strong(expected, true)) { first[dev].store(stream); return TDG_OK; }
  void* f = first[dev].load();
  for (int spin = 0; f == nullptr && stream != nullptr && spin < 1000; ++spin) f = first[dev].load();   // (the claiming thread is between its two stores)
  if (f != stream) {
    tdg_set_error("%s: launched on stream %p, but kernels that take last-block tickets first ran on stream %p of device %d -- their "
                  "device-global ticket slots allow ONE stream per device (a graph that holds such launches must be launched on that "
                  "stream too)", who, stream, f, dev);
    return TDG_EINVAL;
  }
  return TDG_OK;
}
#define TDG_TICKET_STREAM(who, stream) do { const int rc__ = ticket_stream_check(who, stream); if (rc__ != TDG_OK) return rc__; } while (0)

// Two levels above 64 blocks: 4096 tickets on ONE word cost ~46 us (a word takes ~88 atomics per us); 64 sub-counters of
// <= 64 tickets each plus 64 tickets on the top word cost ~1.5 us.
__device__ __forceinline__ bool last_block_ticket(unsigned slot, unsigned sub_base, unsigned nblocks) {
  __shared__ int s_last_t;
  __syncthreads();                                   // every thread of the block is done with what the finisher will change
  if (threadIdx.x == 0) {
    int last = 0;
    if (nblocks <= 64) {
      const unsigned t = __hip_atomic_fetch_add(&g_ticket[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = t == nblocks - 1;
    } else {
      const unsigned sub = blockIdx.x & 63u;
      const unsigned gsize = nblocks / 64u + (sub < (nblocks & 63u) ? 1u : 0u);
      const unsigned t = __hip_atomic_fetch_add(&g_ticket[sub_base + sub], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == gsize - 1) {
        __hip_atomic_store(&g_ticket[sub_base + sub], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned t2 = __hip_atomic_fetch_add(&g_ticket[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = t2 == 63u;
      }
    }
    if (last) __hip_atomic_store(&g_ticket[slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last_t = last;
  }
  __syncthreads();
  return s_last_t != 0;
}

__device__ __forceinline__ bool last_block_arrives(unsigned slot, unsigned nblocks) {
  __shared__ int s_last_a;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(&g_ticket[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = t == nblocks - 1;
    if (last) {
      __hip_atomic_store(&g_ticket[slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    s_last_a = last;
  }
  __syncthreads();
  return s_last_a != 0;
}

// ============================================================================ column reductions
// partial[blk][v][c] = sum over the block's rows of f_v(row, c), v < 2.  2-D grid: blockIdx.x walks
// row blocks, blockIdx.y walks column chunks of CL*VW channels; each lane owns VW consecutive
// channels (one 8/16-byte load per row when VW = 4).
struct ColGeom {
  int rows, C, cs, nblk, rows_per_blk, CL, ncol, vw;  // CL = channel lanes (power of two <= 64)
  int xcs;                                              // row stride of the primary tensor (defaults to cs)
};
static inline ColGeom col_geom(int rows, int C, int cs, const void* p0 = nullptr, const void* p1 = nullptr, int es = 4,
                               bool allow_vec = true) {
  ColGeom g;
  g.rows = rows; g.C = C; g.cs = cs; g.xcs = cs;
  const bool al = (((uintptr_t)p0 | (uintptr_t)p1) % (4 * es)) == 0;
  g.vw = (allow_vec && C % 4 == 0 && cs % 4 == 0 && al) ? 4 : 1;
  const int cv = (C + g.vw - 1) / g.vw;
  int cl = 1;
  while (cl < cv && cl < 64) cl <<= 1;
  g.CL = cl;
  g.ncol = (cv + cl - 1) / cl;
  // wide tensors (>= 8 column chunks: the flattened 4 x 4 x 800 input of the critic's fc2) with few rows: 32-row blocks, or the grid
  // is 1 - 2 workgroups per CU (1024 rows x 12800 columns: 8 x 50 workgroups, 17 us = 1.5 TB/s)
  int nblk = g.ncol >= 8 ? (rows + 31) / 32 : (rows + 127) / 128;
  const int cap = g.ncol >= 8 ? 32 : (g.ncol >= 2 ? 512 : 1024);   // ~4 workgroups per CU for the big single-chunk tensors
  if (nblk > cap) nblk = cap;
  if (nblk < 1) nblk = 1;
  g.rows_per_blk = (rows + nblk - 1) / nblk;
  g.nblk = (rows + g.rows_per_blk - 1) / g.rows_per_blk;
  return g;
}

enum { COL_BN_STATS = 0, COL_BN_BWD = 1, COL_SUM = 2, COL_WSUM = 3 };

struct ColArgs {
  const void* x;        // primary tensor
  const void* y;        // secondary tensor (pre for BN bwd)
  const float* beta;    // BN beta
  const float* coef;    // per-row coefficients (COL_WSUM)
  int act; float leak;
  float* partial;
};

template <typename T, int VW>
__device__ __forceinline__ void load_vw(const T* p, float (&v)[VW]) {
  if constexpr (VW == 1) {
    v[0] = to_f32<T>(p[0]);
  } else if constexpr (sizeof(T) == 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = t[e];
  } else {
    const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)t[e];
  }
}

template <typename T, int MODE, int VW>
__global__ void __launch_bounds__(256) col_partial_kernel(const ColGeom g, const ColArgs a) {
  // f32 tensors (the parity path) are summed in f64: with 32768-row columns whose terms cancel to ~1e-3 of their size (the
  // batch-norm backward sums of the generator at batch 512) f32 running sums were what kept the gradients off the
  // float64 oracle by 2-3e-3; bf16 tensors (the throughput path) keep f32 sums
  using Acc = typename std::conditional<sizeof(T) == 4, double, float>::type;
  __shared__ Acc sh[2][VW][256];
  const int CL = g.CL, RL = 256 / CL;
  const int tx = threadIdx.x % CL, ty = threadIdx.x / CL;
  const int r0 = blockIdx.x * g.rows_per_blk;
  const int r1 = min(g.rows, r0 + g.rows_per_blk);
  // blockIdx.z = group (tdg_bn_fwd_groups: ngroups tensors of g.rows rows behind each other, one set of partial planes each)
  const T* x = static_cast<const T*>(a.x) + (size_t)blockIdx.z * g.rows * g.xcs;
  const T* y = static_cast<const T*>(a.y);
  float* const partial = a.partial + (size_t)blockIdx.z * gridDim.x * 2 * g.C;
  const int c = (blockIdx.y * CL + tx) * VW;
  Acc s0[VW], s1[VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) s0[e] = s1[e] = 0;
  if (c < g.C) {
    float pivot[VW], bta[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) pivot[e] = bta[e] = 0.f;
    if (MODE == COL_BN_STATS) load_vw<T, VW>(x + c, pivot);
    if (MODE == COL_BN_BWD) {
#pragma unroll
      for (int e = 0; e < VW; ++e) bta[e] = a.beta[c + e];
    }
    // four rows per trip, their loads issued before any use (one load in flight per thread left the big
    // single-chunk tensors at 1.3-1.9 TB/s)
    for (int rb = r0 + ty; rb < r1; rb += 4 * RL) {
      float v4[4][VW], p4[4][VW];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = rb + u * RL;
        if (r < r1) {
          load_vw<T, VW>(x + (size_t)r * g.xcs + c, v4[u]);
          if (MODE == COL_BN_BWD) load_vw<T, VW>(y + (size_t)r * g.cs + c, p4[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = rb + u * RL;
        if (r >= r1) continue;
        float (&v)[VW] = v4[u];
        if (MODE == COL_BN_STATS) {
#pragma unroll
          for (int e = 0; e < VW; ++e) { const float d = v[e] - pivot[e]; s0[e] += d; s1[e] += d * d; }
        } else if (MODE == COL_BN_BWD) {
          float (&pre)[VW] = p4[u];
#pragma unroll
          for (int e = 0; e < VW; ++e) {
            const float dpre = v[e] * act_deriv_from_pre(pre[e], a.act, a.leak);
            s0[e] += dpre; s1[e] += dpre * (pre[e] - bta[e]);
          }
        } else if (MODE == COL_SUM) {
#pragma unroll
          for (int e = 0; e < VW; ++e) s0[e] += v[e];
        } else {
          const float k = a.coef ? a.coef[r] : 1.f;
#pragma unroll
          for (int e = 0; e < VW; ++e) s0[e] += v[e] * k;
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < VW; ++e) { sh[0][e][threadIdx.x] = s0[e]; sh[1][e][threadIdx.x] = s1[e]; }
  __syncthreads();
  if (ty == 0 && c < g.C) {
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      Acc t0 = 0, t1 = 0;
      for (int k = 0; k < RL; ++k) { t0 += sh[0][e][k * CL + tx]; t1 += sh[1][e][k * CL + tx]; }
      partial[((size_t)blockIdx.x * 2 + 0) * g.C + c + e] = (float)t0;
      partial[((size_t)blockIdx.x * 2 + 1) * g.C + c + e] = (float)t1;
    }
  }
}

// finalize modes
enum { FIN_BN_STATS = 0, FIN_BN_BWD = 1, FIN_ACC = 2, FIN_ACC2 = 3 };   // ACC2: both planes, out0 / out1
struct FinArgs {
  const float* partial; int nblk, C, rows;
  const void* x0;       // first row of the tensor (pivot) for BN stats
  const float* pivot_f; // ... or an f32 pivot row (the producing conv's bias) when the partials came from its epilogue; both null: 0
  float eps;
  float* out0;          // stats (mean | rstd)   / dbeta / dw
  float* out1;          // bwd: mean sums [2][C] for the apply pass
  float beta_acc;
  long long x0_group;   // FIN_BN_STATS with blockIdx.y = group: elements between the groups' first rows (their partial planes
                        // and [2][C] statistics lie behind each other)
};
// CH channels x 256 / CH partial lanes per block: 8 x 32 for up to ~1k row blocks (C / 8 workgroups, short per-thread chains); 2 x 128
// for the thousands of row tiles a GEMM epilogue reports on pix2pix's 128 x 128 / 256 x 256 layers (the 8 x 32 form: 9 - 20 us on
// 8 - 64 workgroups).  Fixed summation order (deterministic).
#define FIN_CH 8
template <typename T, int MODE, int CH>
__global__ void __launch_bounds__(256) col_finalize_kernel(const FinArgs a) {
  constexpr int RL = 256 / CH;
  using Acc = typename std::conditional<sizeof(T) == 4, double, float>::type;     // (see col_partial_kernel)
  __shared__ Acc sh[2][RL][CH];
  const int tx = threadIdx.x & (CH - 1), ty = threadIdx.x / CH;
  const int c = blockIdx.x * CH + tx;
  const float* const partial = a.partial + (size_t)blockIdx.y * a.nblk * 2 * a.C;
  Acc s0 = 0, s1 = 0;
  if (c < a.C)
    for (int b0 = ty; b0 < a.nblk; b0 += 4 * RL) {
      float p0[4], p1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int b = b0 + RL * u;
        const bool ok = b < a.nblk;
        p0[u] = ok ? partial[((size_t)b * 2 + 0) * a.C + c] : 0.f;
        p1[u] = (ok && MODE != FIN_ACC) ? partial[((size_t)b * 2 + 1) * a.C + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { s0 += p0[u]; s1 += p1[u]; }
    }
  sh[0][ty][tx] = s0;
  sh[1][ty][tx] = s1;
  __syncthreads();
  if (ty != 0 || c >= a.C) return;
  s0 = s1 = 0;
#pragma unroll
  for (int k = 0; k < RL; ++k) { s0 += sh[0][k][tx]; s1 += sh[1][k][tx]; }
  const Acc inv = (Acc)1 / (Acc)a.rows;
  if (MODE == FIN_BN_STATS) {
    const float pivot = a.pivot_f ? a.pivot_f[c] : (a.x0 ? to_f32<T>(static_cast<const T*>(a.x0)[(size_t)blockIdx.y * a.x0_group + c]) : 0.f);
    const Acc md = s0 * inv;
    const Acc var = s1 * inv - md * md > 0 ? s1 * inv - md * md : 0;
    float* const out = a.out0 + (size_t)blockIdx.y * 2 * a.C;
    out[c] = (float)(pivot + md);
    out[a.C + c] = (float)(1.0 / sqrt((double)var + (double)a.eps));
  } else if (MODE == FIN_BN_BWD) {
    a.out0[c] = (float)((a.beta_acc != 0.f ? (Acc)a.beta_acc * a.out0[c] : 0) + s0);
    a.out1[c] = (float)(s0 * inv);
    a.out1[a.C + c] = (float)(s1 * inv);
  } else if (MODE == FIN_ACC2) {
    a.out0[c] = (float)((a.beta_acc != 0.f ? (Acc)a.beta_acc * a.out0[c] : 0) + s0);
    a.out1[c] = (float)((a.beta_acc != 0.f ? (Acc)a.beta_acc * a.out1[c] : 0) + s1);
  } else {
    a.out0[c] = (float)((a.beta_acc != 0.f ? (Acc)a.beta_acc * a.out0[c] : 0) + s0);
  }
}
template <typename T, int MODE>
static void launch_col_finalize(const FinArgs& f, hipStream_t s, int ngroups = 1) {
  if (f.nblk >= 1024)
    hipLaunchKernelGGL((col_finalize_kernel<T, MODE, 2>), dim3((f.C + 1) / 2, ngroups), dim3(256), 0, s, f);
  else
    hipLaunchKernelGGL((col_finalize_kernel<T, MODE, FIN_CH>), dim3((f.C + FIN_CH - 1) / FIN_CH, ngroups), dim3(256), 0, s, f);
}

template <typename T, int VW>
__device__ __forceinline__ void store_vw(T* p, const float (&v)[VW]) {
  if constexpr (VW == 1) {
    p[0] = from_f32<T>(v[0]);
  } else if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  } else {
    *reinterpret_cast<bf16x4*>(p) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  }
}

// lanes: CL channel lanes (VW channels each) x 256/CL row lanes; blockIdx.y = column chunk
template <typename T, int VW>
__global__ void __launch_bounds__(256) bn_fwd_apply_kernel(const ColGeom g, const T* __restrict__ u,
                                                          const float* __restrict__ beta, const float* __restrict__ stats,
                                                          int act, float leak, T* __restrict__ pre, T* __restrict__ h, int hcs) {
  const int CL = g.CL, RL = 256 / CL, C = g.C;
  const int tx = threadIdx.x % CL, ty = threadIdx.x / CL;
  const int c = (blockIdx.y * CL + tx) * VW;
  if (c >= C) return;
  // blockIdx.z = group (tdg_bn_fwd_groups): g.rows rows and one [2][C] set of statistics per group
  stats += (size_t)blockIdx.z * 2 * C;
  u += (size_t)blockIdx.z * g.rows * g.cs;
  if (pre) pre += (size_t)blockIdx.z * g.rows * g.cs;
  h += (size_t)blockIdx.z * g.rows * hcs;
  float mean[VW], rstd[VW], b[VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) { mean[e] = stats[c + e]; rstd[e] = stats[C + c + e]; b[e] = beta[c + e]; }
  for (int r = blockIdx.x * RL + ty; r < g.rows; r += gridDim.x * RL) {
    const size_t i = (size_t)r * g.cs + c;
    float v[VW], p[VW], a[VW];
    load_vw<T, VW>(u + i, v);
#pragma unroll
    for (int e = 0; e < VW; ++e) { p[e] = (v[e] - mean[e]) * rstd[e] + b[e]; a[e] = apply_act(p[e], act, leak); }
    if (pre) store_vw<T, VW>(pre + i, p);                       // (null: a forward pass no backward pass will follow)
    store_vw<T, VW>(h + (size_t)r * hcs + c, a);
  }
}

// DB: also the column sums of the stored du (the bias gradient of the conv in front of the batch norm: analytically zero,
// numerically the same rounding residue a separate pass over du would find) as per-workgroup partials [blk][2][C] plane 0
template <typename T, int VW, bool DB>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const ColGeom g, const T* __restrict__ dh, int dhcs, const T* __restrict__ pre,
                                                          const float* __restrict__ beta, const float* __restrict__ stats,
                                                          const float* __restrict__ sums, int act, float leak, T* __restrict__ du,
                                                          float* __restrict__ db_partial) {
  __shared__ float sh[DB ? VW : 1][DB ? 256 : 1];
  const int CL = g.CL, RL = 256 / CL, C = g.C;
  const int tx = threadIdx.x % CL, ty = threadIdx.x / CL;
  const int c = (blockIdx.y * CL + tx) * VW;
  const bool okc = c < C;
  if (!DB && !okc) return;
  float rstd[VW], b[VW], m0[VW], m1[VW], acc[VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) {
    rstd[e] = okc ? stats[C + c + e] : 0.f; b[e] = okc ? beta[c + e] : 0.f; m0[e] = okc ? sums[c + e] : 0.f; m1[e] = okc ? sums[C + c + e] : 0.f;
    acc[e] = 0.f;
  }
  if (okc)
    for (int r = blockIdx.x * RL + ty; r < g.rows; r += gridDim.x * RL) {
      const size_t i = (size_t)r * g.cs + c;
      float d[VW], p[VW], o[VW];
      load_vw<T, VW>(dh + (size_t)r * dhcs + c, d);
      load_vw<T, VW>(pre + i, p);
#pragma unroll
      for (int e = 0; e < VW; ++e) {
        const float f = act_deriv_from_pre(p[e], act, leak);
        o[e] = rstd[e] * (d[e] * f - m0[e] - (p[e] - b[e]) * m1[e]);
        if (DB) acc[e] += to_f32<T>(from_f32<T>(o[e]));          // the value a pass over the stored tensor would read
      }
      store_vw<T, VW>(du + i, o);
    }
  if constexpr (DB) {
#pragma unroll
    for (int e = 0; e < VW; ++e) sh[e][threadIdx.x] = acc[e];
    __syncthreads();
    if (ty == 0 && okc) {
#pragma unroll
      for (int e = 0; e < VW; ++e) {
        float t = 0.f;
        for (int k = 0; k < RL; ++k) t += sh[e][k * CL + tx];
        db_partial[((size_t)blockIdx.x * 2 + 0) * C + c + e] = t;
      }
    }
  }
}

static inline int apply_row_blocks(const ColGeom& g) {
  const int rl = 256 / g.CL;
  int nb = (g.rows + rl * 8 - 1) / (rl * 8);
  const int cap = 2048 / g.ncol > 1 ? 2048 / g.ncol : 1;
  return nb < 1 ? 1 : (nb > cap ? cap : nb);
}

extern "C" size_t tdg_bn_workspace_bytes(int rows, int c) {
  (void)rows;
  return ((size_t)1024 * 2 * c + 2 * (size_t)c) * sizeof(float);  // upper bound on the row blocks (col_geom's cap)
}
extern "C" size_t tdg_colsum_workspace_bytes(int rows, int cols) { return tdg_bn_workspace_bytes(rows, cols); }

template <typename T, int MODE>
static int run_col_partial(const ColGeom& g, const ColArgs& a, hipStream_t s, int ngroups = 1) {
  if (g.vw == 4)
    hipLaunchKernelGGL((col_partial_kernel<T, MODE, 4>), dim3(g.nblk, g.ncol, ngroups), dim3(256), 0, s, g, a);
  else
    hipLaunchKernelGGL((col_partial_kernel<T, MODE, 1>), dim3(g.nblk, g.ncol, ngroups), dim3(256), 0, s, g, a);
  TDG_HIP_LAUNCH_CHECK("col_partial");
  return TDG_OK;
}

extern "C" int tdg_bn_fwd(int dtype, const void* u, int rows, int c, int cs, const float* beta, float eps, int act,
                          float leak, void* pre, void* h, int h_cs, float* stats, void* workspace, size_t workspace_bytes,
                          void* stream) {
  TDG_CHECK_ARG(u && beta && pre && h && stats && workspace, "tdg_bn_fwd: null pointer");
  TDG_CHECK_ARG(rows > 0 && c > 0 && cs >= c && h_cs >= c, "tdg_bn_fwd: bad shape rows=%d c=%d cs=%d h_cs=%d", rows, c, cs, h_cs);
  if (workspace_bytes < tdg_bn_workspace_bytes(rows, c)) { tdg_set_error("tdg_bn_fwd: workspace too small"); return TDG_EWORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  const ColGeom g = col_geom(rows, c, cs, u, nullptr, tdg_dtype_size(dtype));
  ColArgs a; memset(&a, 0, sizeof(a));
  a.x = u; a.partial = static_cast<float*>(workspace);
  FinArgs f; memset(&f, 0, sizeof(f));
  f.partial = a.partial; f.nblk = g.nblk; f.C = c; f.rows = rows; f.x0 = u; f.eps = eps; f.out0 = stats;
  const ColGeom ga = col_geom(rows, c, cs, u, (const void*)((uintptr_t)pre | (uintptr_t)h), tdg_dtype_size(dtype), h_cs % 4 == 0);
  const dim3 agrid(apply_row_blocks(ga), ga.ncol);
  DISPATCH_T(dtype, {
    int rc = run_col_partial<T, COL_BN_STATS>(g, a, s);
    if (rc) return rc;
    launch_col_finalize<T, FIN_BN_STATS>(f, s);
    if (ga.vw == 4)
      hipLaunchKernelGGL((bn_fwd_apply_kernel<T, 4>), agrid, dim3(256), 0, s, ga, static_cast<const T*>(u), beta, stats, act,
                         leak, static_cast<T*>(pre), static_cast<T*>(h), h_cs);
    else
      hipLaunchKernelGGL((bn_fwd_apply_kernel<T, 1>), agrid, dim3(256), 0, s, ga, static_cast<const T*>(u), beta, stats, act,
                         leak, static_cast<T*>(pre), static_cast<T*>(h), h_cs);
  })
  TDG_HIP_LAUNCH_CHECK("bn_fwd");
  return TDG_OK;
}

// Batch norm + activation of `ngroups` batches that lie behind each other in one tensor, each with ITS OWN batch statistics: the
// generator passes of the n_disc_train critic steps of an iteration run as one pass over n_disc_train x B latent vectors
// (models/gan.py: the generator's variables do not change between them), and ops/layers.py:103,144 normalises per run of the op,
// i.e. per batch of B.  Three launches for all groups (blockIdx.z / .y = group); `pre` may be null (no backward pass follows).
extern "C" int tdg_bn_fwd_groups(int dtype, const void* u, int rows_per_group, int ngroups, int c, int cs, const float* beta,
                                 float eps, int act, float leak, void* pre, void* h, int h_cs, float* stats, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  TDG_CHECK_ARG(u && beta && h && stats && workspace, "tdg_bn_fwd_groups: null pointer");
  TDG_CHECK_ARG(rows_per_group > 0 && ngroups > 0 && ngroups <= 65535 && c > 0 && cs >= c && h_cs >= c,
                "tdg_bn_fwd_groups: bad shape rows=%d groups=%d c=%d cs=%d h_cs=%d", rows_per_group, ngroups, c, cs, h_cs);
  TDG_CHECK_ARG((long long)rows_per_group * ngroups <= 0x7fffffffLL, "tdg_bn_fwd_groups: %d x %d rows", rows_per_group, ngroups);
  TDG_CHECK_ARG(!pre || h_cs == cs || pre != h, "tdg_bn_fwd_groups: pre and h alias with different strides");
  if (workspace_bytes < (size_t)ngroups * tdg_bn_workspace_bytes(rows_per_group, c)) { tdg_set_error("tdg_bn_fwd_groups: workspace too small"); return TDG_EWORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  const int es = tdg_dtype_size(dtype);
  // the groups' first rows must keep the alignment the vector width was chosen for
  const bool grp_al = ((size_t)rows_per_group * cs * es) % (4 * es) == 0 && ((size_t)rows_per_group * h_cs * es) % (4 * es) == 0;
  const ColGeom g = col_geom(rows_per_group, c, cs, u, nullptr, es, grp_al);
  ColArgs a; memset(&a, 0, sizeof(a));
  a.x = u; a.partial = static_cast<float*>(workspace);
  FinArgs f; memset(&f, 0, sizeof(f));
  f.partial = a.partial; f.nblk = g.nblk; f.C = c; f.rows = rows_per_group; f.x0 = u; f.x0_group = (long long)rows_per_group * cs;
  f.eps = eps; f.out0 = stats;
  const ColGeom ga = col_geom(rows_per_group, c, cs, u, (const void*)((uintptr_t)pre | (uintptr_t)h), es, grp_al && h_cs % 4 == 0);
  const dim3 agrid(apply_row_blocks(ga), ga.ncol, ngroups);
  DISPATCH_T(dtype, {
    int rc = run_col_partial<T, COL_BN_STATS>(g, a, s, ngroups);
    if (rc) return rc;
    launch_col_finalize<T, FIN_BN_STATS>(f, s, ngroups);
    if (ga.vw == 4)
      hipLaunchKernelGGL((bn_fwd_apply_kernel<T, 4>), agrid, dim3(256), 0, s, ga, static_cast<const T*>(u), beta, stats, act,
                         leak, static_cast<T*>(pre), static_cast<T*>(h), h_cs);
    else
      hipLaunchKernelGGL((bn_fwd_apply_kernel<T, 1>), agrid, dim3(256), 0, s, ga, static_cast<const T*>(u), beta, stats, act,
                         leak, static_cast<T*>(pre), static_cast<T*>(h), h_cs);
  })
  TDG_HIP_LAUNCH_CHECK("bn_fwd_groups");
  return TDG_OK;
}

extern "C" int tdg_col_finalize_sum(const float* partial, int nblk, int c, float* out, float beta, void* stream) {
  TDG_CHECK_ARG(partial && out && nblk > 0 && c > 0, "tdg_col_finalize_sum: bad argument");
  FinArgs f; memset(&f, 0, sizeof(f));
  f.partial = partial; f.nblk = nblk; f.C = c; f.rows = 1; f.out0 = out; f.beta_acc = beta;
  launch_col_finalize<float, FIN_ACC>(f, (hipStream_t)stream);
  TDG_HIP_LAUNCH_CHECK("col_finalize_sum");
  return TDG_OK;
}

extern "C" int tdg_bn_fwd_from_partials(int dtype, const void* u, int rows, int c, int cs, const float* beta, float eps, int act,
                                        float leak, void* pre, void* h, int h_cs, float* stats, const float* partial, int nblk,
                                        const float* pivot_bias, void* stream) {
  TDG_CHECK_ARG(u && beta && h && stats && partial && nblk > 0, "tdg_bn_fwd_from_partials: null pointer");
  TDG_CHECK_ARG(rows > 0 && c > 0 && cs >= c && h_cs >= c, "tdg_bn_fwd_from_partials: bad shape rows=%d c=%d cs=%d h_cs=%d", rows, c, cs, h_cs);
  hipStream_t s = (hipStream_t)stream;
  FinArgs f; memset(&f, 0, sizeof(f));
  f.partial = partial; f.nblk = nblk; f.C = c; f.rows = rows; f.x0 = nullptr; f.pivot_f = pivot_bias; f.eps = eps; f.out0 = stats;
  const ColGeom ga = col_geom(rows, c, cs, u, (const void*)((uintptr_t)pre | (uintptr_t)h), tdg_dtype_size(dtype), h_cs % 4 == 0);
  const dim3 agrid(apply_row_blocks(ga), ga.ncol);
  DISPATCH_T(dtype, {
    launch_col_finalize<T, FIN_BN_STATS>(f, s);
    if (ga.vw == 4)
      hipLaunchKernelGGL((bn_fwd_apply_kernel<T, 4>), agrid, dim3(256), 0, s, ga, static_cast<const T*>(u), beta, stats, act,
                         leak, static_cast<T*>(pre), static_cast<T*>(h), h_cs);
    else
      hipLaunchKernelGGL((bn_fwd_apply_kernel<T, 1>), agrid, dim3(256), 0, s, ga, static_cast<const T*>(u), beta, stats, act,
                         leak, static_cast<T*>(pre), static_cast<T*>(h), h_cs);
  })
  TDG_HIP_LAUNCH_CHECK("bn_fwd_from_partials");
  return TDG_OK;
}

extern "C" int tdg_bn_bwd(int dtype, const void* dh, int dh_cs, const void* pre, int rows, int c, int cs, const float* beta,
                          const float* stats, int act, float leak, void* du, float* dbeta, float beta_acc,
                          float* dbias, float dbias_acc, void* workspace, size_t workspace_bytes, void* stream) {
  TDG_CHECK_ARG(dh && pre && beta && stats && du && dbeta && workspace, "tdg_bn_bwd: null pointer");
  TDG_CHECK_ARG(rows > 0 && c > 0 && cs >= c && dh_cs >= c, "tdg_bn_bwd: bad shape");
  if (workspace_bytes < tdg_bn_workspace_bytes(rows, c)) { tdg_set_error("tdg_bn_bwd: workspace too small"); return TDG_EWORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  ColGeom g = col_geom(rows, c, cs, dh, pre, tdg_dtype_size(dtype), dh_cs % 4 == 0);
  g.xcs = dh_cs;
  ColArgs a; memset(&a, 0, sizeof(a));
  a.x = dh; a.y = pre; a.beta = beta; a.act = act; a.leak = leak; a.partial = static_cast<float*>(workspace);
  float* sums = a.partial + (size_t)1024 * 2 * c;              // behind the partial planes (which the apply pass reuses)
  FinArgs f; memset(&f, 0, sizeof(f));
  f.partial = a.partial; f.nblk = g.nblk; f.C = c; f.rows = rows; f.out0 = dbeta; f.out1 = sums; f.beta_acc = beta_acc;
  ColGeom ga = col_geom(rows, c, cs, dh, (const void*)((uintptr_t)pre | (uintptr_t)du), tdg_dtype_size(dtype), dh_cs % 4 == 0);
  ga.xcs = dh_cs;
  int arows = apply_row_blocks(ga);
  if (dbias && arows > 1024) arows = 1024;                     // its column partials live in the 1024-block partial planes
  const dim3 agrid(arows, ga.ncol);
  FinArgs fb; memset(&fb, 0, sizeof(fb));
  fb.partial = a.partial; fb.nblk = arows; fb.C = c; fb.rows = rows; fb.out0 = dbias; fb.beta_acc = dbias_acc;
  DISPATCH_T(dtype, {
    int rc = run_col_partial<T, COL_BN_BWD>(g, a, s);
    if (rc) return rc;
    launch_col_finalize<T, FIN_BN_BWD>(f, s);
    if (dbias) {
      if (ga.vw == 4)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 4, true>), agrid, dim3(256), 0, s, ga, static_cast<const T*>(dh), dh_cs,
                           static_cast<const T*>(pre), beta, stats, sums, act, leak, static_cast<T*>(du), a.partial);
      else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 1, true>), agrid, dim3(256), 0, s, ga, static_cast<const T*>(dh), dh_cs,
                           static_cast<const T*>(pre), beta, stats, sums, act, leak, static_cast<T*>(du), a.partial);
      launch_col_finalize<T, FIN_ACC>(fb, s);
    } else {
      if (ga.vw == 4)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 4, false>), agrid, dim3(256), 0, s, ga, static_cast<const T*>(dh), dh_cs,
                           static_cast<const T*>(pre), beta, stats, sums, act, leak, static_cast<T*>(du), nullptr);
      else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T, 1, false>), agrid, dim3(256), 0, s, ga, static_cast<const T*>(dh), dh_cs,
                           static_cast<const T*>(pre), beta, stats, sums, act, leak, static_cast<T*>(du), nullptr);
    }
  })
  TDG_HIP_LAUNCH_CHECK("bn_bwd");
  return TDG_OK;
}

extern "C" int tdg_bias_grad(int dtype, const void* dy, int rows, int c, int cs, float* db, float beta, void* workspace,
                             size_t workspace_bytes, void* stream) {
  TDG_CHECK_ARG(dy && db && workspace && rows > 0 && c > 0 && cs >= c, "tdg_bias_grad: bad argument");
  if (workspace_bytes < tdg_bn_workspace_bytes(rows, c)) { tdg_set_error("tdg_bias_grad: workspace too small"); return TDG_EWORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  const ColGeom g = col_geom(rows, c, cs, dy, nullptr, tdg_dtype_size(dtype));
  ColArgs a; memset(&a, 0, sizeof(a));
  a.x = dy; a.partial = static_cast<float*>(workspace);
  FinArgs f; memset(&f, 0, sizeof(f));
  f.partial = a.partial; f.nblk = g.nblk; f.C = c; f.rows = rows; f.out0 = db; f.beta_acc = beta;
  DISPATCH_T(dtype, {
    int rc = run_col_partial<T, COL_SUM>(g, a, s);
    if (rc) return rc;
    launch_col_finalize<T, FIN_ACC>(f, s);
  })
  TDG_HIP_LAUNCH_CHECK("bias_grad");
  return TDG_OK;
}

extern "C" int tdg_colsum_weighted(int dtype, const void* x, int rows, int cols, int cs, const float* coef, float* dw,
                                   float beta, void* workspace, size_t workspace_bytes, void* stream) {
  TDG_CHECK_ARG(x && dw && workspace && rows > 0 && cols > 0 && cs >= cols, "tdg_colsum_weighted: bad argument");
  if (workspace_bytes < tdg_bn_workspace_bytes(rows, cols)) { tdg_set_error("tdg_colsum_weighted: workspace too small"); return TDG_EWORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  const ColGeom g = col_geom(rows, cols, cs, x, nullptr, tdg_dtype_size(dtype));
  ColArgs a; memset(&a, 0, sizeof(a));
  a.x = x; a.coef = coef; a.partial = static_cast<float*>(workspace);
  FinArgs f; memset(&f, 0, sizeof(f));
  f.partial = a.partial; f.nblk = g.nblk; f.C = cols; f.rows = rows; f.out0 = dw; f.beta_acc = beta;
  DISPATCH_T(dtype, {
    int rc = run_col_partial<T, COL_WSUM>(g, a, s);
    if (rc) return rc;
    launch_col_finalize<T, FIN_ACC>(f, s);
  })
  TDG_HIP_LAUNCH_CHECK("colsum_weighted");
  return TDG_OK;
}

// ============================================================================ instance norm (hem/ops/images.py:73-89)
// y = scale[c] * (x - mu[n,c]) / sqrt(var[n,c] + eps) + shift[c], moments over the H*W positions of ONE image (biased),
// followed by the layer's activation.  One workgroup per (image, 64-channel chunk): 64 channel lanes x 4 row lanes.
// stats[(n*2 + 0)*c + ch] = mu, [(n*2 + 1)*c + ch] = rstd (kept for the backward).
template <typename T>
__global__ void __launch_bounds__(256) instance_norm_fwd_kernel(const T* __restrict__ u, int hw, int c, int cs,
                                                               const float* __restrict__ scale, const float* __restrict__ shift, float eps,
                                                               int act, float leak, T* __restrict__ h, int hcs, float* __restrict__ stats) {
  __shared__ float sh[2][4][64];
  const int n = blockIdx.x, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int ch = blockIdx.y * 64 + tx;
  const bool okc = ch < c;
  const T* un = u + (size_t)n * hw * cs;
  const float pivot = okc ? to_f32<T>(un[ch]) : 0.f;
  float s0 = 0.f, s1 = 0.f;
  if (okc)
    for (int r = ty; r < hw; r += 4) { const float d = to_f32<T>(un[(size_t)r * cs + ch]) - pivot; s0 += d; s1 += d * d; }
  sh[0][ty][tx] = s0; sh[1][ty][tx] = s1;
  __syncthreads();
  s0 = sh[0][0][tx] + sh[0][1][tx] + sh[0][2][tx] + sh[0][3][tx];
  s1 = sh[1][0][tx] + sh[1][1][tx] + sh[1][2][tx] + sh[1][3][tx];
  if (!okc) return;
  const float md = s0 / (float)hw, var = fmaxf(s1 / (float)hw - md * md, 0.f);
  const float mu = pivot + md, rstd = rsqrtf(var + eps);
  if (ty == 0) { stats[((size_t)n * 2 + 0) * c + ch] = mu; stats[((size_t)n * 2 + 1) * c + ch] = rstd; }
  const float g = scale[ch], b = shift[ch];
  T* hn = h + (size_t)n * hw * hcs;
  for (int r = ty; r < hw; r += 4)
    hn[(size_t)r * hcs + ch] = from_f32<T>(apply_act(g * (to_f32<T>(un[(size_t)r * cs + ch]) - mu) * rstd + b, act, leak));
}
extern "C" int tdg_instance_norm_fwd(int dtype, const void* u, int n, int hw, int c, int cs, const float* scale, const float* shift,
                                     float eps, int act, float leak, void* h, int h_cs, float* stats, void* stream) {
  TDG_CHECK_ARG(u && scale && shift && h && stats && n > 0 && hw > 0 && c > 0 && cs >= c && h_cs >= c, "tdg_instance_norm_fwd: bad argument");
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(instance_norm_fwd_kernel<T>, dim3(n, (c + 63) / 64), dim3(256), 0, (hipStream_t)stream, static_cast<const T*>(u),
                       hw, c, cs, scale, shift, eps, act, leak, static_cast<T*>(h), h_cs, stats);
  })
  TDG_HIP_LAUNCH_CHECK("instance_norm_fwd");
  return TDG_OK;
}
// backward: g = dh * act'(pre), pre = scale * xhat + shift;  du = scale * rstd * (g - mean(g) - xhat * mean(g * xhat));
// per-image partials part[(n*2 + 0)*c + ch] = sum g * xhat (dscale), [(n*2 + 1)*c + ch] = sum g (dshift)
template <typename T>
__global__ void __launch_bounds__(256) instance_norm_bwd_kernel(const T* __restrict__ dh, int dhcs, const T* __restrict__ u, int hw, int c,
                                                               int cs, const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ stats, int act, float leak, T* __restrict__ du,
                                                               float* __restrict__ part) {
  __shared__ float sh[2][4][64];
  const int n = blockIdx.x, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int ch = blockIdx.y * 64 + tx;
  const bool okc = ch < c;
  const T* un = u + (size_t)n * hw * cs;
  const T* dn = dh + (size_t)n * hw * dhcs;
  const float mu = okc ? stats[((size_t)n * 2 + 0) * c + ch] : 0.f, rstd = okc ? stats[((size_t)n * 2 + 1) * c + ch] : 0.f;
  const float g0 = okc ? scale[ch] : 0.f, b0 = okc ? shift[ch] : 0.f;
  float s0 = 0.f, s1 = 0.f;
  if (okc)
    for (int r = ty; r < hw; r += 4) {
      const float xh = (to_f32<T>(un[(size_t)r * cs + ch]) - mu) * rstd;
      const float g = to_f32<T>(dn[(size_t)r * dhcs + ch]) * act_deriv_from_pre(g0 * xh + b0, act, leak);
      s0 += g; s1 += g * xh;
    }
  sh[0][ty][tx] = s0; sh[1][ty][tx] = s1;
  __syncthreads();
  s0 = sh[0][0][tx] + sh[0][1][tx] + sh[0][2][tx] + sh[0][3][tx];
  s1 = sh[1][0][tx] + sh[1][1][tx] + sh[1][2][tx] + sh[1][3][tx];
  if (!okc) return;
  if (ty == 0) { part[((size_t)n * 2 + 0) * c + ch] = s1; part[((size_t)n * 2 + 1) * c + ch] = s0; }
  const float m0 = s0 / (float)hw, m1 = s1 / (float)hw;
  T* dun = du + (size_t)n * hw * cs;
  for (int r = ty; r < hw; r += 4) {
    const float xh = (to_f32<T>(un[(size_t)r * cs + ch]) - mu) * rstd;
    const float g = to_f32<T>(dn[(size_t)r * dhcs + ch]) * act_deriv_from_pre(g0 * xh + b0, act, leak);
    dun[(size_t)r * cs + ch] = from_f32<T>(g0 * rstd * (g - m0 - xh * m1));
  }
}
extern "C" int tdg_instance_norm_bwd(int dtype, const void* dh, int dh_cs, const void* u, int n, int hw, int c, int cs,
                                     const float* scale, const float* shift, const float* stats, int act, float leak, void* du,
                                     float* dscale, float* dshift, float beta, void* workspace, size_t workspace_bytes, void* stream) {
  TDG_CHECK_ARG(dh && u && scale && shift && stats && du && dscale && dshift && workspace && n > 0 && hw > 0 && c > 0 && cs >= c && dh_cs >= c,
                "tdg_instance_norm_bwd: bad argument");
  if (workspace_bytes < (size_t)n * 2 * c * sizeof(float)) { tdg_set_error("tdg_instance_norm_bwd: workspace too small"); return TDG_EWORKSPACE; }
  float* part = static_cast<float*>(workspace);
  FinArgs f; memset(&f, 0, sizeof(f));
  f.partial = part; f.nblk = n; f.C = c; f.rows = 1; f.out0 = dscale; f.out1 = dshift; f.beta_acc = beta;
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(instance_norm_bwd_kernel<T>, dim3(n, (c + 63) / 64), dim3(256), 0, (hipStream_t)stream, static_cast<const T*>(dh),
                       dh_cs, static_cast<const T*>(u), hw, c, cs, scale, shift, stats, act, leak, static_cast<T*>(du), part);
    launch_col_finalize<T, FIN_ACC2>(f, (hipStream_t)stream);
  })
  TDG_HIP_LAUNCH_CHECK("instance_norm_bwd");
  return TDG_OK;
}

// out = act(a + b) on flat buffers of one layout (the residual sum of hem/ops/layers.py:215-320; act NONE: plain add)
template <typename T>
__global__ void __launch_bounds__(256) add_act_kernel(const T* __restrict__ a, const T* __restrict__ b, size_t n, int act, float leak,
                                                     T* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    out[i] = from_f32<T>(apply_act(to_f32<T>(a[i]) + to_f32<T>(b[i]), act, leak));
}
extern "C" int tdg_add_act(int dtype, const void* a, const void* b, size_t n, int act, float leak, void* out, void* stream) {
  TDG_CHECK_ARG(a && b && out && n > 0, "tdg_add_act: bad argument");
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(add_act_kernel<T>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, static_cast<const T*>(a),
                       static_cast<const T*>(b), n, act, leak, static_cast<T*>(out));
  })
  TDG_HIP_LAUNCH_CHECK("add_act");
  return TDG_OK;
}

// ============================================================================ row ops (fc2)
// VW elements (16 bytes) of a row per access: bf16 8, f32 4
template <typename T> struct RowVec;
template <> struct RowVec<bf16_t> {
  static constexpr int VW = 8;
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
    *reinterpret_cast<bf16x8*>(p) = bf16x8{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3], (bf16_t)v[4], (bf16_t)v[5], (bf16_t)v[6], (bf16_t)v[7]};
  }
};
template <> struct RowVec<float> {
  static constexpr int VW = 4;
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = t[e];
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) { *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]}; }
};

template <typename T, bool VEC>
__global__ void __launch_bounds__(256) rowdot_kernel(const T* __restrict__ x, int rows, int cols, const float* __restrict__ w,
                                                    const float* __restrict__ bias, int act, float* __restrict__ out) {
  __shared__ float sh[4];
  constexpr int VW = RowVec<T>::VW;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const T* xr = x + (size_t)r * cols;
    float s = 0.f;
    if constexpr (VEC) {
      for (int c = threadIdx.x * VW; c < cols; c += 256 * VW) {
        float v[VW], wv[VW];
        RowVec<T>::load(xr + c, v);
#pragma unroll
        for (int e = 0; e < VW; e += 4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(w + c + e);
          wv[e] = t[0]; wv[e + 1] = t[1]; wv[e + 2] = t[2]; wv[e + 3] = t[3];
        }
#pragma unroll
        for (int e = 0; e < VW; ++e) s += v[e] * wv[e];
      }
    } else {
      for (int c = threadIdx.x; c < cols; c += 256) s += to_f32<T>(xr[c]) * w[c];
    }
    s = block_sum256(s, sh);
    if (threadIdx.x == 0) out[r] = apply_act(s + (bias ? bias[0] : 0.f), act, 0.f);
  }
}

extern "C" int tdg_rowdot(int dtype, const void* x, int rows, int cols, const float* w, const float* bias, int act,
                          float* out, void* stream) {
  TDG_CHECK_ARG(x && w && out && rows > 0 && cols > 0, "tdg_rowdot: bad argument");
  DISPATCH_T(dtype, {
    const bool vec = cols % RowVec<T>::VW == 0 && (((uintptr_t)x | (uintptr_t)w) & 15) == 0;
    if (vec)
      hipLaunchKernelGGL((rowdot_kernel<T, true>), dim3(rows < 4096 ? rows : 4096), dim3(256), 0, (hipStream_t)stream,
                         static_cast<const T*>(x), rows, cols, w, bias, act, out);
    else
      hipLaunchKernelGGL((rowdot_kernel<T, false>), dim3(rows < 4096 ? rows : 4096), dim3(256), 0, (hipStream_t)stream,
                         static_cast<const T*>(x), rows, cols, w, bias, act, out);
  })
  TDG_HIP_LAUNCH_CHECK("rowdot");
  return TDG_OK;
}

template <typename T, bool VEC>
__global__ void __launch_bounds__(256) rowouter_kernel(const float* __restrict__ dout, const float* __restrict__ w, int rows,
                                                      int cols, int mask_mode, float leak, const T* __restrict__ msk,
                                                      T* __restrict__ dx) {
  constexpr int VW = RowVec<T>::VW;
  const float mlow = mask_low(mask_mode, leak);
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const float d = dout[r];
    if constexpr (VEC) {
      for (int c = threadIdx.x * VW; c < cols; c += 256 * VW) {
        const size_t i = (size_t)r * cols + c;
        float v[VW], m[VW];
#pragma unroll
        for (int e = 0; e < VW; e += 4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(w + c + e);
          v[e] = d * t[0]; v[e + 1] = d * t[1]; v[e + 2] = d * t[2]; v[e + 3] = d * t[3];
        }
        if (mask_mode != TDG_MASK_NONE) {
          RowVec<T>::load(msk + i, m);
#pragma unroll
          for (int e = 0; e < VW; ++e) v[e] *= m[e] > 0.f ? 1.f : mlow;
        }
        RowVec<T>::store(dx + i, v);
      }
    } else {
      for (int c = threadIdx.x; c < cols; c += 256) {
        const size_t i = (size_t)r * cols + c;
        float v = d * w[c];
        if (mask_mode != TDG_MASK_NONE) v *= mask_factor(to_f32<T>(msk[i]), mask_mode, leak);
        dx[i] = from_f32<T>(v);
      }
    }
  }
}

extern "C" int tdg_rowouter(int dtype, const float* dout, const float* w, int rows, int cols, int mask_mode, float leak,
                            const void* mask_src, void* dx, void* stream) {
  TDG_CHECK_ARG(dout && w && dx && rows > 0 && cols > 0, "tdg_rowouter: bad argument");
  TDG_CHECK_ARG(mask_mode == TDG_MASK_NONE || mask_src, "tdg_rowouter: mask without mask_src");
  DISPATCH_T(dtype, {
    const bool vec = cols % RowVec<T>::VW == 0 && (((uintptr_t)w | (uintptr_t)mask_src | (uintptr_t)dx) & 15) == 0;
    if (vec)
      hipLaunchKernelGGL((rowouter_kernel<T, true>), dim3(rows < 4096 ? rows : 4096), dim3(256), 0, (hipStream_t)stream, dout, w,
                         rows, cols, mask_mode, leak, static_cast<const T*>(mask_src), static_cast<T*>(dx));
    else
      hipLaunchKernelGGL((rowouter_kernel<T, false>), dim3(rows < 4096 ? rows : 4096), dim3(256), 0, (hipStream_t)stream, dout, w,
                         rows, cols, mask_mode, leak, static_cast<const T*>(mask_src), static_cast<T*>(dx));
  })
  TDG_HIP_LAUNCH_CHECK("rowouter");
  return TDG_OK;
}

// ============================================================================ flat elementwise
template <typename T>
__global__ void __launch_bounds__(256) bias_act_kernel(const T* __restrict__ x, int rows, int C, int cs,
                                                      const float* __restrict__ bias, int act, float leak, T* __restrict__ y) {
  for (int r = blockIdx.x; r < rows; r += gridDim.x)
    for (int c = threadIdx.x; c < C; c += 256) {
      const size_t i = (size_t)r * cs + c;
      y[i] = from_f32<T>(apply_act(to_f32<T>(x[i]) + (bias ? bias[c] : 0.f), act, leak));
    }
}
extern "C" int tdg_bias_act(int dtype, const void* x, int rows, int c, int cs, const float* bias, int act, float leak,
                            void* y, void* stream) {
  TDG_CHECK_ARG(x && y && rows > 0 && c > 0 && cs >= c, "tdg_bias_act: bad argument");
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(bias_act_kernel<T>, dim3(rows < 4096 ? rows : 4096), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const T*>(x), rows, c, cs, bias, act, leak, static_cast<T*>(y));
  })
  TDG_HIP_LAUNCH_CHECK("bias_act");
  return TDG_OK;
}

template <typename T>
__global__ void __launch_bounds__(256) act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ post, size_t n, int act,
                                                     float leak, T* __restrict__ dx) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float p = to_f32<T>(post[i]);
    float f = 1.f;
    if (act == TDG_ACT_TANH) f = 1.f - p * p;
    else if (act == TDG_ACT_SIGMOID) f = p * (1.f - p);
    else if (act == TDG_ACT_LRELU) f = p > 0.f ? 1.f : leak;
    else if (act == TDG_ACT_RELU) f = p > 0.f ? 1.f : 0.f;
    dx[i] = from_f32<T>(to_f32<T>(dy[i]) * f);
  }
}
extern "C" int tdg_act_bwd(int dtype, const void* dy, const void* post, size_t n, int act, float leak, void* dx,
                           void* stream) {
  TDG_CHECK_ARG(dy && post && dx && n > 0, "tdg_act_bwd: bad argument");
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(act_bwd_kernel<T>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, static_cast<const T*>(dy),
                       static_cast<const T*>(post), n, act, leak, static_cast<T*>(dx));
  })
  TDG_HIP_LAUNCH_CHECK("act_bwd");
  return TDG_OK;
}

template <typename T>
__global__ void __launch_bounds__(256) affine_cast_kernel(const float* __restrict__ in, size_t n, float scale, float shift,
                                                         T* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    out[i] = from_f32<T>(scale * (in[i] + shift));
}
extern "C" int tdg_affine_cast(int dtype, const float* in, size_t n, float scale, float shift, void* out, void* stream) {
  TDG_CHECK_ARG(in && out && n > 0, "tdg_affine_cast: bad argument");
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(affine_cast_kernel<T>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, in, n, scale, shift,
                       static_cast<T*>(out));
  })
  TDG_HIP_LAUNCH_CHECK("affine_cast");
  return TDG_OK;
}
template <typename T>
__global__ void __launch_bounds__(256) affine_cast_rows_kernel(const float* __restrict__ in, size_t n, int c, int cs, float scale,
                                                              float shift, T* __restrict__ out) {
  if (c <= 16) {              // images: one pixel per thread and trip, no 64-bit division per element
    const size_t rows = n / (size_t)c;
    for (size_t r = blockIdx.x * (size_t)256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256)
      for (int ch = 0; ch < c; ++ch) out[r * cs + ch] = from_f32<T>(scale * (in[r * c + ch] + shift));
    return;
  }
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / c;
    const int ch = (int)(i - r * c);
    out[r * cs + ch] = from_f32<T>(scale * (in[i] + shift));
  }
}
// pix2pix's critic input: out_ab[r] = scale * ([a[r] | b[r]] + shift), out_a0[r] = scale * ([a[r] | .] + shift) with zeros in b's
// channels (the generator writes them later), ca + cb == 4 channels per pixel: one pass with 8 / 16-byte stores instead of three
// tdg_affine_cast_rows passes with 2-byte ones
template <typename T>
__global__ void __launch_bounds__(256) affine_cast_pair_kernel(const float* __restrict__ a, int ca, const float* __restrict__ b, int cb,
                                                              size_t rows, float scale, float shift, T* __restrict__ out_ab,
                                                              T* __restrict__ out_a0) {
  for (size_t r = blockIdx.x * (size_t)256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256) {
    float v[4] = {0.f, 0.f, 0.f, 0.f}, w[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < ca; ++ch) v[ch] = w[ch] = scale * (a[r * ca + ch] + shift);
    for (int ch = 0; ch < cb; ++ch) v[ca + ch] = scale * (b[r * cb + ch] + shift);
    store_vw<T, 4>(out_ab + r * 4, v);
    store_vw<T, 4>(out_a0 + r * 4, w);
  }
}
extern "C" int tdg_affine_cast_pair(int dtype, const float* a, int ca, const float* b, int cb, size_t rows, float scale, float shift,
                                    void* out_ab, void* out_a0, void* stream) {
  TDG_CHECK_ARG(a && b && out_ab && out_a0 && rows > 0 && ca > 0 && cb > 0 && ca + cb == 4, "tdg_affine_cast_pair: bad argument");
  TDG_CHECK_ARG((((uintptr_t)out_ab | (uintptr_t)out_a0) & 15) == 0, "tdg_affine_cast_pair: outputs must be 16-byte aligned");
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(affine_cast_pair_kernel<T>, dim3(ew_blocks(rows)), dim3(256), 0, (hipStream_t)stream, a, ca, b, cb, rows, scale,
                       shift, static_cast<T*>(out_ab), static_cast<T*>(out_a0));
  })
  TDG_HIP_LAUNCH_CHECK("affine_cast_pair");
  return TDG_OK;
}

extern "C" int tdg_affine_cast_rows(int dtype, const float* in, int rows, int c, int cs, float scale, float shift, void* out,
                                    void* stream) {
  TDG_CHECK_ARG(in && out && rows > 0 && c > 0 && cs >= c, "tdg_affine_cast_rows: bad argument");
  const size_t n = (size_t)rows * c;
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(affine_cast_rows_kernel<T>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, in, n, c, cs, scale,
                       shift, static_cast<T*>(out));
  })
  TDG_HIP_LAUNCH_CHECK("affine_cast_rows");
  return TDG_OK;
}
extern "C" int tdg_cast_from_f32(int dtype, const float* in, size_t n, void* out, void* stream) {
  return tdg_affine_cast(dtype, in, n, 1.f, 0.f, out, stream);
}

template <typename T>
__global__ void __launch_bounds__(256) cast_to_f32_kernel(const T* __restrict__ in, size_t n, float* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = to_f32<T>(in[i]);
}
extern "C" int tdg_cast_to_f32(int dtype, const void* in, size_t n, float* out, void* stream) {
  TDG_CHECK_ARG(in && out && n > 0, "tdg_cast_to_f32: bad argument");
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(cast_to_f32_kernel<T>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, static_cast<const T*>(in),
                       n, out);
  })
  TDG_HIP_LAUNCH_CHECK("cast_to_f32");
  return TDG_OK;
}

template <typename T>
__global__ void __launch_bounds__(256) gp_interp_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                       const float* __restrict__ alpha, int rows, int cols, T* __restrict__ xhat) {
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const float al = alpha[r];
    for (int c = threadIdx.x; c < cols; c += 256) {
      const size_t i = (size_t)r * cols + c;
      const float xv = to_f32<T>(x[i]);
      xhat[i] = from_f32<T>(xv + al * (to_f32<T>(g[i]) - xv));
    }
  }
}
extern "C" int tdg_gp_interp(int dtype, const void* x, const void* g, const float* alpha, int rows, int cols, void* xhat,
                             void* stream) {
  TDG_CHECK_ARG(x && g && alpha && xhat && rows > 0 && cols > 0, "tdg_gp_interp: bad argument");
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(gp_interp_kernel<T>, dim3(rows < 4096 ? rows : 4096), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const T*>(x), static_cast<const T*>(g), alpha, rows, cols, static_cast<T*>(xhat));
  })
  TDG_HIP_LAUNCH_CHECK("gp_interp");
  return TDG_OK;
}

// ---- scalar reductions -----------------------------------------------------------------------
#define RED_BLOCKS 256
extern "C" size_t tdg_reduce_workspace_bytes(size_t) { return RED_BLOCKS * sizeof(float); }

// acc = beta * acc + sum x^2: per-block partials, summed by the block that arrives last in a fixed order (deterministic)
template <typename T>
__global__ void __launch_bounds__(256) sumsq_partial_kernel(const T* __restrict__ x, size_t n, float* __restrict__ partial,
                                                           float* __restrict__ acc, float beta, float gp_lambda = 0.f,
                                                           float* __restrict__ gp_scal = nullptr) {
  __shared__ float sh[4];
  float s = 0.f;
  constexpr int VW = 16 / (int)sizeof(T);
  if ((((uintptr_t)x) & 15) == 0) {                    // 16 bytes per access (scalar 2-byte loads ran at 0.4 TB/s)
    const size_t nv = n / VW;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
      float v[VW];
      if constexpr (sizeof(T) == 2) {
        const bf16x8 t = reinterpret_cast<const bf16x8*>(x)[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
      } else {
        const f32x4 t = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = t[e];
      }
#pragma unroll
      for (int e = 0; e < VW; ++e) s += v[e] * v[e];
    }
    for (size_t i = nv * VW + blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
      const float v = to_f32<T>(x[i]);
      s += v * v;
    }
  } else {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
      const float v = to_f32<T>(x[i]);
      s += v * v;
    }
  }
  s = block_sum256(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
  if (!last_block_arrives(TICKET_SUMSQ, gridDim.x)) return;
  float t = 0.f;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) t += partial[i];
  t = block_sum256(t, sh);
  if (threadIdx.x == 0) {
    const float ss = (beta != 0.f ? beta * acc[0] : 0.f) + t;
    acc[0] = ss;
    if (gp_scal) {                                          // tdg_gp_sumsq: the penalty scalars of gp_scalars_kernel from the same launch
      const float sl = sqrtf(ss);
      gp_scal[0] = (sl - 1.f) * (sl - 1.f);
      gp_scal[1] = gp_lambda * 2.f * (sl - 1.f) / sl;
    }
  }
}
// out[seg] = beta * out[seg] + scale * sum(x[seg * seglen ...]) : one block per segment
__global__ void __launch_bounds__(256) reduce_final_kernel(const float* __restrict__ partial, int np, float* __restrict__ acc,
                                                          float beta, float scale) {
  __shared__ float sh[4];
  const float* p = partial + (size_t)blockIdx.x * np;
  float s = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) s += p[i];
  s = block_sum256(s, sh);
  if (threadIdx.x == 0) acc[blockIdx.x] = (beta != 0.f ? beta * acc[blockIdx.x] : 0.f) + scale * s;
}
extern "C" int tdg_sumsq(int dtype, const void* x, size_t n, float* acc, float beta, void* workspace,
                         size_t workspace_bytes, void* stream) {
  TDG_CHECK_ARG(x && acc && workspace && n > 0, "tdg_sumsq: bad argument");
  if (workspace_bytes < RED_BLOCKS * sizeof(float)) { tdg_set_error("tdg_sumsq: workspace too small"); return TDG_EWORKSPACE; }
  TDG_TICKET_STREAM("tdg_sumsq", stream);
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(sumsq_partial_kernel<T>, dim3(RED_BLOCKS), dim3(256), 0, (hipStream_t)stream, static_cast<const T*>(x),
                       n, static_cast<float*>(workspace), acc, beta);
  })
  TDG_HIP_LAUNCH_CHECK("sumsq");
  return TDG_OK;
}
// tdg_sumsq(beta = 0) + tdg_gp_scalars in one launch: sumsq[0] = sum x^2, scal[0] = (sqrt(sumsq) - 1)^2, scal[1] = lambda * 2 (s - 1) / s
extern "C" int tdg_gp_sumsq(int dtype, const void* x, size_t n, float* sumsq, float lambda, float* scal, void* workspace,
                            size_t workspace_bytes, void* stream) {
  TDG_CHECK_ARG(x && sumsq && scal && workspace && n > 0, "tdg_gp_sumsq: bad argument");
  if (workspace_bytes < RED_BLOCKS * sizeof(float)) { tdg_set_error("tdg_gp_sumsq: workspace too small"); return TDG_EWORKSPACE; }
  TDG_TICKET_STREAM("tdg_gp_sumsq", stream);
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(sumsq_partial_kernel<T>, dim3(RED_BLOCKS), dim3(256), 0, (hipStream_t)stream, static_cast<const T*>(x),
                       n, static_cast<float*>(workspace), sumsq, 0.f, lambda, scal);
  })
  TDG_HIP_LAUNCH_CHECK("gp_sumsq");
  return TDG_OK;
}
extern "C" int tdg_mean_segments_f32(const float* x, int nseg, int seglen, float* out, void* stream) {
  TDG_CHECK_ARG(x && out && nseg > 0 && nseg <= 65535 && seglen > 0, "tdg_mean_segments_f32: bad argument");
  hipLaunchKernelGGL(reduce_final_kernel, dim3(nseg), dim3(256), 0, (hipStream_t)stream, x, seglen, out, 0.f, 1.f / (float)seglen);
  TDG_HIP_LAUNCH_CHECK("mean_segments_f32");
  return TDG_OK;
}
// out[0] = beta * out[0] + sum x  (one block: the bias gradient of a one-column dense layer = the sum of its seeds)
extern "C" int tdg_sum_f32(const float* x, int n, float* out, float beta, void* stream) {
  TDG_CHECK_ARG(x && out && n > 0, "tdg_sum_f32: bad argument");
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, out, beta, 1.f);
  TDG_HIP_LAUNCH_CHECK("sum_f32");
  return TDG_OK;
}
extern "C" int tdg_mean_f32(const float* x, int n, float* out, void* stream) {
  TDG_CHECK_ARG(x && out && n > 0, "tdg_mean_f32: bad argument");
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, out, 0.f, 1.f / (float)n);
  TDG_HIP_LAUNCH_CHECK("mean_f32");
  return TDG_OK;
}

__global__ void __launch_bounds__(256) gan_logloss_kernel(const float* __restrict__ dr, const float* __restrict__ df, int n,
                                                         float* __restrict__ sr, float* __restrict__ sfd, float* __restrict__ sfg,
                                                         float* __restrict__ scal) {
  __shared__ float sh[4];
  float ld = 0.f, lg = 0.f;
  const float inv = 1.f / (float)n;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float r = dr[i], f = df[i];
    ld += -logf(r + 1e-8f) - logf(1.f - f + 1e-8f);
    lg += -logf(f + 1e-8f);
    sr[i] = -inv / (r + 1e-8f) * r * (1.f - r);
    sfd[i] = inv / (1.f - f + 1e-8f) * f * (1.f - f);
    sfg[i] = -inv / (f + 1e-8f) * f * (1.f - f);
  }
  ld = block_sum256(ld, sh);
  lg = block_sum256(lg, sh);
  if (threadIdx.x == 0) { scal[0] = ld * inv; scal[1] = lg * inv; }
}
extern "C" int tdg_gan_logloss(const float* d_real, const float* d_fake, int n, float* seed_real, float* seed_fake_d,
                               float* seed_fake_g, float* scal, void* stream) {
  TDG_CHECK_ARG(d_real && d_fake && seed_real && seed_fake_d && seed_fake_g && scal && n > 0, "tdg_gan_logloss: bad argument");
  hipLaunchKernelGGL(gan_logloss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d_real, d_fake, n, seed_real, seed_fake_d,
                     seed_fake_g, scal);
  TDG_HIP_LAUNCH_CHECK("gan_logloss");
  return TDG_OK;
}

// ---- pix2pix losses ----------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) p2p_xent_kernel(const T* __restrict__ z, int rows, int cs, int mode, T* __restrict__ seed,
                                                      float* __restrict__ scal) {
  __shared__ float sh[4];
  float dr = 0.f, df = 0.f, gf = 0.f;
  const float inv = 1.f / (float)rows;
  for (int i = threadIdx.x; i < rows; i += 256) {
    const float zr = to_f32<T>(z[(size_t)i * cs]), zf = to_f32<T>(z[(size_t)(rows + i) * cs]);
    const float sr = log1pf(expf(-fabsf(zr))), sf = log1pf(expf(-fabsf(zf)));
    dr += fmaxf(zr, 0.f) - zr + sr;
    df += fmaxf(zf, 0.f) + sf;
    gf += fmaxf(zf, 0.f) - zf + sf;
    if (mode) {
      const float pr = 1.f / (1.f + expf(-zr)), pf = 1.f / (1.f + expf(-zf));
      seed[(size_t)i * cs] = from_f32<T>(mode == 1 ? (pr - 1.f) * inv : 0.f);
      seed[(size_t)(rows + i) * cs] = from_f32<T>(mode == 1 ? pf * inv : (pf - 1.f) * inv);
    }
  }
  dr = block_sum256(dr, sh);
  df = block_sum256(df, sh);
  gf = block_sum256(gf, sh);
  if (threadIdx.x == 0) { scal[0] = dr * inv; scal[1] = df * inv; scal[2] = gf * inv; }
}
extern "C" int tdg_p2p_xent(int dtype, const void* logits, int rows, int cs, int mode, void* seed, float* scal, void* stream) {
  TDG_CHECK_ARG(logits && scal && rows > 0 && cs > 0 && (mode == 0 || seed), "tdg_p2p_xent: bad argument");
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(p2p_xent_kernel<T>, dim3(1), dim3(256), 0, (hipStream_t)stream, static_cast<const T*>(logits), rows, cs,
                       mode, static_cast<T*>(seed), scal);
  })
  TDG_HIP_LAUNCH_CHECK("p2p_xent");
  return TDG_OK;
}
template <typename T>
__global__ void __launch_bounds__(256) p2p_l1_kernel(const T* __restrict__ y, const T* __restrict__ g, int rows, int cs, float wgt,
                                                    T* __restrict__ dg, int dgs, float* __restrict__ partial) {
  __shared__ float sh[4];
  float a = 0.f, q = 0.f;
  const float inv = 1.f / (float)rows;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rows; i += gridDim.x * 256) {
    const float d = 0.5f * (to_f32<T>(g[(size_t)i * cs]) - to_f32<T>(y[(size_t)i * cs]));   // g01 - y01
    a += fabsf(d);
    q += d * d;
    if (dg) {
      const float s = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      dg[(size_t)i * dgs] = from_f32<T>(to_f32<T>(dg[(size_t)i * dgs]) + wgt * 0.5f * s * inv);
    }
  }
  a = block_sum256(a, sh);
  q = block_sum256(q, sh);
  if (threadIdx.x == 0) { partial[blockIdx.x] = a; partial[RED_BLOCKS + blockIdx.x] = q; }
}
__global__ void __launch_bounds__(256) p2p_l1_final_kernel(const float* __restrict__ partial, int rows, float* __restrict__ scal) {
  __shared__ float sh[4];
  float a = 0.f, q = 0.f;
  for (int i = threadIdx.x; i < RED_BLOCKS; i += 256) { a += partial[i]; q += partial[RED_BLOCKS + i]; }
  a = block_sum256(a, sh);
  q = block_sum256(q, sh);
  if (threadIdx.x == 0) { scal[0] = a / (float)rows; scal[1] = sqrtf(q / (float)rows); }
}
extern "C" int tdg_p2p_l1(int dtype, const void* y, const void* g, int rows, int cs, float weight, void* dg, int dgs, float* scal,
                          void* workspace, size_t workspace_bytes, void* stream) {
  TDG_CHECK_ARG(y && g && scal && workspace && rows > 0 && cs > 0, "tdg_p2p_l1: bad argument");
  if (workspace_bytes < 2 * RED_BLOCKS * sizeof(float)) { tdg_set_error("tdg_p2p_l1: workspace too small"); return TDG_EWORKSPACE; }
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(p2p_l1_kernel<T>, dim3(RED_BLOCKS), dim3(256), 0, (hipStream_t)stream, static_cast<const T*>(y),
                       static_cast<const T*>(g), rows, cs, weight, static_cast<T*>(dg), dgs, static_cast<float*>(workspace));
  })
  hipLaunchKernelGGL(p2p_l1_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, static_cast<const float*>(workspace), rows, scal);
  TDG_HIP_LAUNCH_CHECK("p2p_l1");
  return TDG_OK;
}

// ---- VAE ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) vae_reparam_kernel(const T* __restrict__ heads, int hs, const T* __restrict__ eps, int es,
                                                         int rows, int L, T* __restrict__ z, int zs) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rows * L; i += gridDim.x * 256) {
    const int r = i / L, c = i - r * L;
    z[(size_t)r * zs + c] = from_f32<T>(to_f32<T>(heads[(size_t)r * hs + c]) +
                                       to_f32<T>(heads[(size_t)r * hs + L + c]) * to_f32<T>(eps[(size_t)r * es + c]));
  }
}
extern "C" int tdg_vae_reparam(int dtype, const void* heads, int hs, const void* eps, int es, int rows, int L, void* z, int zs,
                               void* stream) {
  TDG_CHECK_ARG(heads && eps && z && rows > 0 && L > 0 && hs >= 2 * L && es >= L && zs >= L, "tdg_vae_reparam: bad argument");
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(vae_reparam_kernel<T>, dim3(ew_blocks((size_t)rows * L, 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const T*>(heads), hs, static_cast<const T*>(eps), es, rows, L, static_cast<T*>(z), zs);
  })
  TDG_HIP_LAUNCH_CHECK("vae_reparam");
  return TDG_OK;
}
template <typename T>
__global__ void __launch_bounds__(256) vae_reparam_bwd_kernel(const T* __restrict__ dz, int zs, const T* __restrict__ eps, int es,
                                                             int rows, int L, T* __restrict__ dh, int hs,
                                                             const T* __restrict__ heads, int hsin, float klw) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rows * L; i += gridDim.x * 256) {
    const int r = i / L, c = i - r * L;
    const float g = to_f32<T>(dz[(size_t)r * zs + c]);
    float gm = g, gs = g * to_f32<T>(eps[(size_t)r * es + c]);
    if (heads) {            // + klw * d latent_loss / d (mean, stddev), latent_loss = 0.5 sum(m^2 + s^2 - log(1e-8 + s^2) - 1)
      const float m = to_f32<T>(heads[(size_t)r * hsin + c]), sd = to_f32<T>(heads[(size_t)r * hsin + L + c]);
      gm += klw * m;
      gs += klw * (sd - sd / (1e-8f + sd * sd));
    }
    dh[(size_t)r * hs + c] = from_f32<T>(gm);
    dh[(size_t)r * hs + L + c] = from_f32<T>(gs);
  }
}
extern "C" int tdg_vae_reparam_bwd(int dtype, const void* dz, int zs, const void* eps, int es, int rows, int L, void* dheads,
                                   int hs, void* stream) {
  TDG_CHECK_ARG(dz && eps && dheads && rows > 0 && L > 0 && hs >= 2 * L && es >= L && zs >= L, "tdg_vae_reparam_bwd: bad argument");
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(vae_reparam_bwd_kernel<T>, dim3(ew_blocks((size_t)rows * L, 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const T*>(dz), zs, static_cast<const T*>(eps), es, rows, L, static_cast<T*>(dheads), hs,
                       (const T*)nullptr, 0, 0.f);
  })
  TDG_HIP_LAUNCH_CHECK("vae_reparam_bwd");
  return TDG_OK;
}
extern "C" int tdg_vae_reparam_bwd_kl(int dtype, const void* dz, int zs, const void* eps, int es, const void* heads, int hs_in,
                                      float kl_weight, int rows, int L, void* dheads, int hs, void* stream) {
  TDG_CHECK_ARG(dz && eps && heads && dheads && rows > 0 && L > 0 && hs >= 2 * L && hs_in >= 2 * L, "tdg_vae_reparam_bwd_kl: bad argument");
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(vae_reparam_bwd_kernel<T>, dim3(ew_blocks((size_t)rows * L, 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const T*>(dz), zs, static_cast<const T*>(eps), es, rows, L, static_cast<T*>(dheads), hs,
                       static_cast<const T*>(heads), hs_in, kl_weight);
  })
  TDG_HIP_LAUNCH_CHECK("vae_reparam_bwd_kl");
  return TDG_OK;
}
template <typename T>
__global__ void __launch_bounds__(256) vae_kl_partial_kernel(const T* __restrict__ heads, int hs, int rows, int L,
                                                            float* __restrict__ partial) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rows * L; i += gridDim.x * 256) {
    const int r = i / L, c = i - r * L;
    const float m = to_f32<T>(heads[(size_t)r * hs + c]), sd = to_f32<T>(heads[(size_t)r * hs + L + c]);
    s += m * m + sd * sd - logf(1e-8f + sd * sd) - 1.f;
  }
  s = block_sum256(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
extern "C" int tdg_vae_kl(int dtype, const void* heads, int hs, int rows, int L, float* scal, void* workspace,
                          size_t workspace_bytes, void* stream) {
  TDG_CHECK_ARG(heads && scal && workspace && rows > 0 && L > 0 && hs >= 2 * L, "tdg_vae_kl: bad argument");
  if (workspace_bytes < RED_BLOCKS * sizeof(float)) { tdg_set_error("tdg_vae_kl: workspace too small"); return TDG_EWORKSPACE; }
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(vae_kl_partial_kernel<T>, dim3(RED_BLOCKS), dim3(256), 0, (hipStream_t)stream, static_cast<const T*>(heads),
                       hs, rows, L, static_cast<float*>(workspace));
  })
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, static_cast<const float*>(workspace),
                     RED_BLOCKS, scal, 0.f, 0.5f);
  TDG_HIP_LAUNCH_CHECK("vae_kl");
  return TDG_OK;
}
template <typename T>
__global__ void __launch_bounds__(256) vae_bce_kernel(const float* __restrict__ x, const T* __restrict__ d, size_t n, int c, int cs,
                                                     T* __restrict__ seed, float* __restrict__ partial) {
  __shared__ float sh[4];
  float s = 0.f;
  // one pixel row per thread and trip (an element-indexed loop paid a 64-bit division per element: 69 us for 6.3 M elements)
  const size_t rows = n / (size_t)c;
  for (size_t r = blockIdx.x * (size_t)256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256) {
    const float* xr = x + r * c;
    const T* dr = d + r * cs;
    T* sr = seed + r * cs;
    for (int ch = 0; ch < c; ++ch) {
      const float xv = xr[ch], dv = to_f32<T>(dr[ch]);
      s -= xv * logf(1e-8f + dv) + (1.f - xv) * logf(1e-8f + (1.f - dv));
      sr[ch] = from_f32<T>(-(xv / (1e-8f + dv) - (1.f - xv) / (1e-8f + (1.f - dv))));
    }
  }
  s = block_sum256(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
extern "C" int tdg_vae_bce(int dtype, const float* x, const void* d, int rows, int c, int cs, void* seed, float* scal,
                           void* workspace, size_t workspace_bytes, void* stream) {
  TDG_CHECK_ARG(x && d && seed && scal && workspace && rows > 0 && c > 0 && cs >= c, "tdg_vae_bce: bad argument");
  if (workspace_bytes < RED_BLOCKS * sizeof(float)) { tdg_set_error("tdg_vae_bce: workspace too small"); return TDG_EWORKSPACE; }
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(vae_bce_kernel<T>, dim3(RED_BLOCKS), dim3(256), 0, (hipStream_t)stream, x, static_cast<const T*>(d),
                       (size_t)rows * c, c, cs, static_cast<T*>(seed), static_cast<float*>(workspace));
  })
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, static_cast<const float*>(workspace),
                     RED_BLOCKS, scal, 0.f, 1.f);
  TDG_HIP_LAUNCH_CHECK("vae_bce");
  return TDG_OK;
}

template <typename T>
__global__ void __launch_bounds__(256) l1_loss_kernel(const float* __restrict__ x, const T* __restrict__ d, size_t n, int c, int cs,
                                                     float scale, float shift, float inv_n, T* __restrict__ seed,
                                                     float* __restrict__ partial) {
  __shared__ float sh[4];
  float s = 0.f;
  const size_t rows = n / (size_t)c;
  for (size_t r = blockIdx.x * (size_t)256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256) {
    const float* xr = x + r * c;
    const T* dr = d + r * cs;
    T* sr = seed + r * cs;
    for (int ch = 0; ch < c; ++ch) {
      const float u = scale * (xr[ch] + shift) - to_f32<T>(dr[ch]);      // x - d on the rescaled input
      s += fabsf(u);
      sr[ch] = from_f32<T>(u > 0.f ? -inv_n : (u < 0.f ? inv_n : 0.f));   // AbsGrad: sign(u), sign(0) = 0; d u / d d = -1
    }
  }
  s = block_sum256(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
extern "C" int tdg_l1_loss(int dtype, const float* x, const void* d, int rows, int c, int cs, float scale, float shift, void* seed,
                           float* scal, void* workspace, size_t workspace_bytes, void* stream) {
  TDG_CHECK_ARG(x && d && seed && scal && workspace && rows > 0 && c > 0 && cs >= c, "tdg_l1_loss: bad argument");
  if (workspace_bytes < RED_BLOCKS * sizeof(float)) { tdg_set_error("tdg_l1_loss: workspace too small"); return TDG_EWORKSPACE; }
  const float inv_n = 1.f / ((float)rows * (float)c);
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(l1_loss_kernel<T>, dim3(RED_BLOCKS), dim3(256), 0, (hipStream_t)stream, x, static_cast<const T*>(d),
                       (size_t)rows * c, c, cs, scale, shift, inv_n, static_cast<T*>(seed), static_cast<float*>(workspace));
  })
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, static_cast<const float*>(workspace),
                     RED_BLOCKS, scal, 0.f, inv_n);
  TDG_HIP_LAUNCH_CHECK("l1_loss");
  return TDG_OK;
}

// tf.nn.dropout(x, keep_prob) given the uniform draws u: x * floor(keep_prob + u) / keep_prob, in place.  The same call
// with the same u on the incoming gradient is its backward.
template <typename T>
__global__ void __launch_bounds__(256) dropout_kernel(T* __restrict__ y, size_t n, int c, int ycs, const float* __restrict__ u,
                                                     float keep, float inv_keep) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / c;
    const size_t j = r * ycs + (i - r * c);
    y[j] = from_f32<T>(to_f32<T>(y[j]) * floorf(keep + u[i]) * inv_keep);
  }
}
extern "C" int tdg_dropout(int dtype, void* y, int rows, int c, int ycs, const float* u, float keep, void* stream) {
  TDG_CHECK_ARG(y && u && rows > 0 && c > 0 && ycs >= c && keep > 0.f && keep <= 1.f, "tdg_dropout: bad argument");
  const size_t n = (size_t)rows * c;
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(dropout_kernel<T>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, static_cast<T*>(y), n, c, ycs, u,
                       keep, 1.f / keep);
  })
  TDG_HIP_LAUNCH_CHECK("dropout");
  return TDG_OK;
}

__global__ void gp_scalars_kernel(const float* __restrict__ ss, float lambda, float* __restrict__ scal) {
  const float s = sqrtf(ss[0]);
  scal[0] = (s - 1.f) * (s - 1.f);
  scal[1] = lambda * 2.f * (s - 1.f) / s;
}
extern "C" int tdg_gp_scalars(const float* sumsq, float lambda, float* scal, void* stream) {
  TDG_CHECK_ARG(sumsq && scal, "tdg_gp_scalars: null pointer");
  hipLaunchKernelGGL(gp_scalars_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, lambda, scal);
  TDG_HIP_LAUNCH_CHECK("gp_scalars");
  return TDG_OK;
}

template <typename T>
__global__ void __launch_bounds__(256) scale_by_dev_kernel(const T* __restrict__ in, size_t n, const float* __restrict__ coef,
                                                          T* __restrict__ out) {
  const float k = coef[0];
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    out[i] = from_f32<T>(k * to_f32<T>(in[i]));
}
extern "C" int tdg_scale_by_dev(int dtype, const void* in, size_t n, const float* coef, void* out, void* stream) {
  TDG_CHECK_ARG(in && coef && out && n > 0, "tdg_scale_by_dev: bad argument");
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(scale_by_dev_kernel<T>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, static_cast<const T*>(in),
                       n, coef, static_cast<T*>(out));
  })
  TDG_HIP_LAUNCH_CHECK("scale_by_dev");
  return TDG_OK;
}

// ---- --gp_per_sample (SURVEY App. C-4 opt-in): per-image norms instead of one norm over the whole batch tensor.
// One block per row: ss = sum v^2, slopes = sqrt(ss), pen[r] = (slopes - 1)^2, coef[r] = lambda * 2 (slopes - 1) / (slopes * rows)
// (d (lambda * mean_r pen_r) / d v_r = coef[r] * v_r), and u = coef[r] * v written in the same pass over the row.
template <typename T>
__global__ void __launch_bounds__(256) gp_rows_kernel(const T* __restrict__ v, int rows, int cols, float lambda,
                                                     float* __restrict__ pen, T* __restrict__ u) {
  __shared__ float sh[4];
  __shared__ float s_coef;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const T* vr = v + (size_t)r * cols;
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) { const float x = to_f32<T>(vr[c]); s += x * x; }
    s = block_sum256(s, sh);
    if (threadIdx.x == 0) {
      const float sl = sqrtf(s);
      pen[r] = (sl - 1.f) * (sl - 1.f);
      s_coef = lambda * 2.f * (sl - 1.f) / (sl * (float)rows);
    }
    __syncthreads();
    const float k = s_coef;
    for (int c = threadIdx.x; c < cols; c += 256) u[(size_t)r * cols + c] = from_f32<T>(k * to_f32<T>(vr[c]));
    __syncthreads();
  }
}
extern "C" int tdg_gp_rows(int dtype, const void* v, int rows, int cols, float lambda, float* pen_rows, void* u, void* stream) {
  TDG_CHECK_ARG(v && pen_rows && u && rows > 0 && cols > 0, "tdg_gp_rows: bad argument");
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(gp_rows_kernel<T>, dim3(rows < 4096 ? rows : 4096), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const T*>(v), rows, cols, lambda, pen_rows, static_cast<T*>(u));
  })
  TDG_HIP_LAUNCH_CHECK("gp_rows");
  return TDG_OK;
}

__global__ void __launch_bounds__(256) fill_kernel(float* __restrict__ x, size_t n, float v) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) x[i] = v;
}
extern "C" int tdg_fill_f32(float* x, size_t n, float value, void* stream) {
  TDG_CHECK_ARG(x && n > 0, "tdg_fill_f32: bad argument");
  hipLaunchKernelGGL(fill_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, n, value);
  TDG_HIP_LAUNCH_CHECK("fill");
  return TDG_OK;
}

// ============================================================================ optimizers
// 16-byte vectors; buckets are padded to a multiple of 4 floats by the host.
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, size_t n4, float lr_t, float b1, float b2, float eps,
                                                  float gs) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 gv = gs * reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i], pv = reinterpret_cast<f32x4*>(p)[i];
    mv = b1 * mv + (1.f - b1) * gv;
    vv = b2 * vv + (1.f - b2) * gv * gv;
#pragma unroll
    for (int e = 0; e < 4; ++e) pv[e] -= lr_t * mv[e] / (sqrtf(vv[e]) + eps);
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    reinterpret_cast<f32x4*>(p)[i] = pv;
  }
}
extern "C" int tdg_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float beta1, float beta2,
                             float eps, float grad_scale, void* stream) {
  TDG_CHECK_ARG(p && g && m && v && n > 0 && (n & 3) == 0, "tdg_adam_step: bad argument (n must be a multiple of 4)");
  hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n / 4, 512)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n / 4, lr_t,
                     beta1, beta2, eps, grad_scale);
  TDG_HIP_LAUNCH_CHECK("adam");
  return TDG_OK;
}

__global__ void __launch_bounds__(256) adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                      float* __restrict__ v, size_t n4, float lr, float b1, float b2, float eps,
                                                      float gs, int* __restrict__ t_dev) {
  const float t = (float)(t_dev[0] + 1);
  const float lr_t = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 gv = gs * reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i], pv = reinterpret_cast<f32x4*>(p)[i];
    mv = b1 * mv + (1.f - b1) * gv;
    vv = b2 * vv + (1.f - b2) * gv * gv;
#pragma unroll
    for (int e = 0; e < 4; ++e) pv[e] -= lr_t * mv[e] / (sqrtf(vv[e]) + eps);
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    reinterpret_cast<f32x4*>(p)[i] = pv;
  }
  // every block has read the step count by the time it takes its ticket: the last one counts the step
  if (last_block_ticket(TICKET_ADAM, TICKET_ADAM_SUB, gridDim.x) && threadIdx.x == 0) t_dev[0] += 1;
}
extern "C" int tdg_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                                 float eps, float grad_scale, int32_t* t_dev, void* stream) {
  TDG_CHECK_ARG(p && g && m && v && t_dev && n > 0 && (n & 3) == 0, "tdg_adam_step_dev: bad argument (n must be a multiple of 4)");
  TDG_TICKET_STREAM("tdg_adam_step_dev", stream);
  hipLaunchKernelGGL(adam_dev_kernel, dim3(ew_blocks(n / 4, 512)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n / 4, lr,
                     beta1, beta2, eps, grad_scale, t_dev);
  TDG_HIP_LAUNCH_CHECK("adam_dev");
  return TDG_OK;
}
__global__ void add_i32_kernel(int* x, int inc) { x[0] += inc; }
extern "C" int tdg_add_i32(int32_t* x, int32_t inc, void* stream) {
  TDG_CHECK_ARG(x, "tdg_add_i32: null pointer");
  hipLaunchKernelGGL(add_i32_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, x, inc);
  TDG_HIP_LAUNCH_CHECK("add_i32");
  return TDG_OK;
}

__global__ void __launch_bounds__(256) rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ rms,
                                                     float* __restrict__ mom, size_t n4, float lr, float decay, float mu,
                                                     float eps, float gs) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 gv = gs * reinterpret_cast<const f32x4*>(g)[i];
    f32x4 rv = reinterpret_cast<f32x4*>(rms)[i], mv = reinterpret_cast<f32x4*>(mom)[i], pv = reinterpret_cast<f32x4*>(p)[i];
    rv = decay * rv + (1.f - decay) * gv * gv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mv[e] = mu * mv[e] + lr * gv[e] / sqrtf(rv[e] + eps);
      pv[e] -= mv[e];
    }
    reinterpret_cast<f32x4*>(rms)[i] = rv;
    reinterpret_cast<f32x4*>(mom)[i] = mv;
    reinterpret_cast<f32x4*>(p)[i] = pv;
  }
}
extern "C" int tdg_rmsprop_step(float* p, const float* g, float* rms, float* mom, size_t n, float lr, float decay,
                                float momentum, float eps, float grad_scale, void* stream) {
  TDG_CHECK_ARG(p && g && rms && mom && n > 0 && (n & 3) == 0, "tdg_rmsprop_step: bad argument");
  hipLaunchKernelGGL(rmsprop_kernel, dim3(ew_blocks(n / 4, 512)), dim3(256), 0, (hipStream_t)stream, p, g, rms, mom, n / 4,
                     lr, decay, momentum, eps, grad_scale);
  TDG_HIP_LAUNCH_CHECK("rmsprop");
  return TDG_OK;
}

// RMSProp with centered=True: the running mean of g is subtracted from the second moment (mg slot starts at 0).
__global__ void __launch_bounds__(256) rmsprop_centered_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                              float* __restrict__ mg, float* __restrict__ rms,
                                                              float* __restrict__ mom, size_t n4, float lr, float decay,
                                                              float mu, float eps, float gs) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 gv = gs * reinterpret_cast<const f32x4*>(g)[i];
    f32x4 rv = reinterpret_cast<f32x4*>(rms)[i], mv = reinterpret_cast<f32x4*>(mom)[i], pv = reinterpret_cast<f32x4*>(p)[i];
    f32x4 av = reinterpret_cast<f32x4*>(mg)[i];
    rv = decay * rv + (1.f - decay) * gv * gv;
    av = decay * av + (1.f - decay) * gv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mv[e] = mu * mv[e] + lr * gv[e] / sqrtf(rv[e] - av[e] * av[e] + eps);
      pv[e] -= mv[e];
    }
    reinterpret_cast<f32x4*>(rms)[i] = rv;
    reinterpret_cast<f32x4*>(mg)[i] = av;
    reinterpret_cast<f32x4*>(mom)[i] = mv;
    reinterpret_cast<f32x4*>(p)[i] = pv;
  }
}
extern "C" int tdg_rmsprop_centered_step(float* p, const float* g, float* mg, float* rms, float* mom, size_t n, float lr,
                                         float decay, float momentum, float eps, float grad_scale, void* stream) {
  TDG_CHECK_ARG(p && g && mg && rms && mom && n > 0 && (n & 3) == 0, "tdg_rmsprop_centered_step: bad argument");
  hipLaunchKernelGGL(rmsprop_centered_kernel, dim3(ew_blocks(n / 4, 512)), dim3(256), 0, (hipStream_t)stream, p, g, mg, rms,
                     mom, n / 4, lr, decay, momentum, eps, grad_scale);
  TDG_HIP_LAUNCH_CHECK("rmsprop_centered");
  return TDG_OK;
}

// Adagrad: acc += g^2; p -= lr * g / sqrt(acc)   (acc slot initialised by the caller, 0.1 in TF)
__global__ void __launch_bounds__(256) adagrad_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                     float* __restrict__ acc, size_t n4, float lr, float gs) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 gv = gs * reinterpret_cast<const f32x4*>(g)[i];
    f32x4 av = reinterpret_cast<f32x4*>(acc)[i] + gv * gv, pv = reinterpret_cast<f32x4*>(p)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) pv[e] -= lr * gv[e] / sqrtf(av[e]);
    reinterpret_cast<f32x4*>(acc)[i] = av;
    reinterpret_cast<f32x4*>(p)[i] = pv;
  }
}
extern "C" int tdg_adagrad_step(float* p, const float* g, float* acc, size_t n, float lr, float grad_scale, void* stream) {
  TDG_CHECK_ARG(p && g && acc && n > 0 && (n & 3) == 0, "tdg_adagrad_step: bad argument");
  hipLaunchKernelGGL(adagrad_kernel, dim3(ew_blocks(n / 4, 512)), dim3(256), 0, (hipStream_t)stream, p, g, acc, n / 4, lr,
                     grad_scale);
  TDG_HIP_LAUNCH_CHECK("adagrad");
  return TDG_OK;
}

// Adadelta: acc = rho*acc + (1-rho) g^2; u = sqrt(acc_u + eps) / sqrt(acc + eps) * g; acc_u = rho*acc_u + (1-rho) u^2;
// p -= lr * u
__global__ void __launch_bounds__(256) adadelta_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ acc, float* __restrict__ acc_u, size_t n4,
                                                      float lr, float rho, float eps, float gs) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 gv = gs * reinterpret_cast<const f32x4*>(g)[i];
    f32x4 av = reinterpret_cast<f32x4*>(acc)[i], uv = reinterpret_cast<f32x4*>(acc_u)[i], pv = reinterpret_cast<f32x4*>(p)[i];
    av = rho * av + (1.f - rho) * gv * gv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float u = sqrtf(uv[e] + eps) / sqrtf(av[e] + eps) * gv[e];
      uv[e] = rho * uv[e] + (1.f - rho) * u * u;
      pv[e] -= lr * u;
    }
    reinterpret_cast<f32x4*>(acc)[i] = av;
    reinterpret_cast<f32x4*>(acc_u)[i] = uv;
    reinterpret_cast<f32x4*>(p)[i] = pv;
  }
}
extern "C" int tdg_adadelta_step(float* p, const float* g, float* acc, float* acc_update, size_t n, float lr, float rho,
                                 float eps, float grad_scale, void* stream) {
  TDG_CHECK_ARG(p && g && acc && acc_update && n > 0 && (n & 3) == 0, "tdg_adadelta_step: bad argument");
  hipLaunchKernelGGL(adadelta_kernel, dim3(ew_blocks(n / 4, 512)), dim3(256), 0, (hipStream_t)stream, p, g, acc, acc_update,
                     n / 4, lr, rho, eps, grad_scale);
  TDG_HIP_LAUNCH_CHECK("adadelta");
  return TDG_OK;
}

// FTRL-proximal with lr_power = -0.5 and the l1 / l2 strengths TF defaults to zero:
//   acc' = acc + g^2; lin += g - (sqrt(acc') - sqrt(acc)) / lr * p; quad = sqrt(acc') / lr + 2 l2;
//   p = |lin| > l1 ? (sign(lin) l1 - lin) / quad : 0
__global__ void __launch_bounds__(256) ftrl_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ acc, float* __restrict__ lin, size_t n4, float lr,
                                                  float l1, float l2, float gs) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const f32x4 gv = gs * reinterpret_cast<const f32x4*>(g)[i];
    f32x4 av = reinterpret_cast<f32x4*>(acc)[i], lv = reinterpret_cast<f32x4*>(lin)[i], pv = reinterpret_cast<f32x4*>(p)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float an = av[e] + gv[e] * gv[e];
      const float sn = sqrtf(an), so = sqrtf(av[e]);
      lv[e] += gv[e] - (sn - so) / lr * pv[e];
      const float quad = sn / lr + 2.f * l2;
      const float sgn = lv[e] > 0.f ? 1.f : (lv[e] < 0.f ? -1.f : 0.f);
      pv[e] = fabsf(lv[e]) > l1 ? (sgn * l1 - lv[e]) / quad : 0.f;
      av[e] = an;
    }
    reinterpret_cast<f32x4*>(acc)[i] = av;
    reinterpret_cast<f32x4*>(lin)[i] = lv;
    reinterpret_cast<f32x4*>(p)[i] = pv;
  }
}
extern "C" int tdg_ftrl_step(float* p, const float* g, float* acc, float* linear, size_t n, float lr, float l1, float l2,
                             float grad_scale, void* stream) {
  TDG_CHECK_ARG(p && g && acc && linear && n > 0 && (n & 3) == 0 && lr > 0.f, "tdg_ftrl_step: bad argument");
  hipLaunchKernelGGL(ftrl_kernel, dim3(ew_blocks(n / 4, 512)), dim3(256), 0, (hipStream_t)stream, p, g, acc, linear, n / 4,
                     lr, l1, l2, grad_scale);
  TDG_HIP_LAUNCH_CHECK("ftrl");
  return TDG_OK;
}

__global__ void __launch_bounds__(256) sgdm_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ acc,
                                                  size_t n4, float lr, float mu, float gs) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 a = mu * reinterpret_cast<f32x4*>(acc)[i] + gs * reinterpret_cast<const f32x4*>(g)[i];
    reinterpret_cast<f32x4*>(acc)[i] = a;
    reinterpret_cast<f32x4*>(p)[i] -= lr * a;
  }
}
extern "C" int tdg_sgd_momentum_step(float* p, const float* g, float* acc, size_t n, float lr, float momentum,
                                     float grad_scale, void* stream) {
  TDG_CHECK_ARG(p && g && acc && n > 0 && (n & 3) == 0, "tdg_sgd_momentum_step: bad argument");
  hipLaunchKernelGGL(sgdm_kernel, dim3(ew_blocks(n / 4, 512)), dim3(256), 0, (hipStream_t)stream, p, g, acc, n / 4, lr,
                     momentum, grad_scale);
  TDG_HIP_LAUNCH_CHECK("sgd_momentum");
  return TDG_OK;
}

__global__ void __launch_bounds__(256) clamp_kernel(float* __restrict__ p, size_t n, float lo, float hi) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = fminf(fmaxf(p[i], lo), hi);
}
extern "C" int tdg_clamp(float* p, size_t n, float lo, float hi, void* stream) {
  TDG_CHECK_ARG(p && n > 0, "tdg_clamp: bad argument");
  hipLaunchKernelGGL(clamp_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, n, lo, hi);
  TDG_HIP_LAUNCH_CHECK("clamp");
  return TDG_OK;
}

__global__ void __launch_bounds__(256) check_finite_kernel(const float* __restrict__ x, size_t n, int* __restrict__ flag) {
  int bad = 0;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) bad |= !isfinite(x[i]);
  if (__any(bad) && (threadIdx.x & 63) == 0) flag[0] = 1;   // racing identical stores are benign
}
extern "C" int tdg_check_finite(const float* x, size_t n, int* flag, void* stream) {
  TDG_CHECK_ARG(x && flag && n > 0, "tdg_check_finite: bad argument");
  hipLaunchKernelGGL(check_finite_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, n, flag);
  TDG_HIP_LAUNCH_CHECK("check_finite");
  return TDG_OK;
}

// ============================================================================ Philox4x32-10
struct Philox {
  uint32_t c[4], k[2];
  __device__ __forceinline__ void round() {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
  }
  __device__ __forceinline__ void gen(uint64_t seed, uint64_t stream_id, uint64_t ctr) {
    c[0] = (uint32_t)ctr; c[1] = (uint32_t)(ctr >> 32); c[2] = (uint32_t)stream_id; c[3] = (uint32_t)(stream_id >> 32);
    k[0] = (uint32_t)seed; k[1] = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) round();
  }
};
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }   // [0,1)

template <typename T>
__global__ void __launch_bounds__(256) random_normal_kernel(uint64_t seed, uint64_t sid, uint64_t offset, int* __restrict__ draw_dev,
                                                           size_t n, T* __restrict__ out) {
  if (draw_dev) offset = ((uint64_t)(unsigned)(draw_dev[0] + 1)) << 24;
  const size_t n4 = (n + 3) >> 2;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    Philox ph;
    ph.gen(seed, sid, offset + i);
    float z[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float u1 = 1.0f - u01(ph.c[2 * h]);     // (0,1]
      const float u2 = u01(ph.c[2 * h + 1]);
      const float r = sqrtf(-2.f * logf(u1));
      z[2 * h] = r * cosf(6.283185307179586f * u2);
      z[2 * h + 1] = r * sinf(6.283185307179586f * u2);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * i + e < n) out[4 * i + e] = from_f32<T>(z[e]);
  }
  // device-counter form: the draw is counted by the block that finishes last (every block read the counter at its start)
  if (draw_dev && last_block_ticket(TICKET_RNG, TICKET_RNG_SUB, gridDim.x) && threadIdx.x == 0) draw_dev[0] += 1;
}
extern "C" int tdg_random_normal(int dtype, uint64_t seed, uint64_t stream_id, uint64_t offset, size_t n, void* out,
                                 void* stream) {
  TDG_CHECK_ARG(out && n > 0, "tdg_random_normal: bad argument");
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(random_normal_kernel<T>, dim3(ew_blocks((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, seed,
                       stream_id, offset, (int*)nullptr, n, static_cast<T*>(out));
  })
  TDG_HIP_LAUNCH_CHECK("random_normal");
  return TDG_OK;
}
extern "C" int tdg_random_normal_dev(int dtype, uint64_t seed, uint64_t stream_id, int32_t* draw_dev, size_t n, void* out,
                                     void* stream) {
  TDG_CHECK_ARG(out && draw_dev && n > 0, "tdg_random_normal_dev: bad argument");
  TDG_TICKET_STREAM("tdg_random_normal_dev", stream);
  DISPATCH_T(dtype, {
    hipLaunchKernelGGL(random_normal_kernel<T>, dim3(ew_blocks((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, seed,
                       stream_id, (uint64_t)0, draw_dev, n, static_cast<T*>(out));
  })
  TDG_HIP_LAUNCH_CHECK("random_normal_dev");
  return TDG_OK;
}

__global__ void __launch_bounds__(256) random_uniform_kernel(uint64_t seed, uint64_t sid, uint64_t offset,
                                                            int* __restrict__ draw_dev, size_t n, float* __restrict__ out) {
  if (draw_dev) offset = ((uint64_t)(unsigned)(draw_dev[0] + 1)) << 24;
  const size_t n4 = (n + 3) >> 2;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    Philox ph;
    ph.gen(seed, sid, offset + i);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * i + e < n) out[4 * i + e] = u01(ph.c[e]);
  }
  if (draw_dev && last_block_ticket(TICKET_RNG, TICKET_RNG_SUB, gridDim.x) && threadIdx.x == 0) draw_dev[0] += 1;
}
extern "C" int tdg_random_uniform_f32(uint64_t seed, uint64_t stream_id, uint64_t offset, size_t n, float* out, void* stream) {
  TDG_CHECK_ARG(out && n > 0, "tdg_random_uniform_f32: bad argument");
  hipLaunchKernelGGL(random_uniform_kernel, dim3(ew_blocks((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, seed,
                     stream_id, offset, (int*)nullptr, n, out);
  TDG_HIP_LAUNCH_CHECK("random_uniform");
  return TDG_OK;
}
extern "C" int tdg_random_uniform_f32_dev(uint64_t seed, uint64_t stream_id, int32_t* draw_dev, size_t n, float* out,
                                          void* stream) {
  TDG_CHECK_ARG(out && draw_dev && n > 0, "tdg_random_uniform_f32_dev: bad argument");
  TDG_TICKET_STREAM("tdg_random_uniform_f32_dev", stream);
  hipLaunchKernelGGL(random_uniform_kernel, dim3(ew_blocks((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, seed,
                     stream_id, (uint64_t)0, draw_dev, n, out);
  TDG_HIP_LAUNCH_CHECK("random_uniform_dev");
  return TDG_OK;
}
