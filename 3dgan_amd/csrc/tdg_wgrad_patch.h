// Patch-resident filter-gradient GEMM (tdg_wgrad_patch.hip): plan + launch, called from bwd_filter_impl (tdg_igemm.hip).
#pragma once
#include "tdg_igemm.h"

// The gathered operand of a 64-row step (64 / (GH*GW) whole images) is staged ONCE as a patch of source pixels
// [image][row][column][nsl 8-channel slices] instead of as 64 im2col rows: a 5x5 stride-2 filter re-reads every source pixel
// for 6.25 taps, and the CU's L2 -> LDS intake is what bounds the slab form of this GEMM.
struct WpPlan {
  int nsl;        // 16-byte chunks (8-channel slices) per patch pixel: the slices a 32-unit tile of K can span
  int rp;         // patch row pitch in chunks (>= SW * nsl; the padding is chosen against bank conflicts)
  int imgs;       // images per step
  int npa;        // 1 KiB LDS-DMA pieces per patch
  int nslices;    // C / 8
  int nunits;     // nslices * ntaps: K in (slice, tap) units of 8 filter rows; a tile holds 32 of them
  int nslot;      // pieces per wave and step
  int conflicts;  // simulated extra LDS cycles of the chosen layout (diagnostics)
  FastDiv fd_nt, fd_rp, fd_nsl;
};
bool tdg_wgrad_patch_plan(const WgArgs& a, WpPlan* p);
int tdg_wgrad_patch_launch(WgArgs& a, WpPlan& p, double flops, hipStream_t s);
