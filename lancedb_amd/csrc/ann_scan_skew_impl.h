// ann_scan_skew_impl.h — launcher template shared by the two translation units that instantiate k_scan_skew
// (ann_scan_skew.hip: the plain widths; ann_scan_skew_slab.hip: the padded / multi-slab form), so that the two
// kernel families compile in parallel.
#pragma once
#include "ann_internal.h"
#include "kernels_ivfpq.h"
#include "kernels_skew.h"

// IMG: the kernels that copy a pre-built table image (SkewArgs::lut_img, kernels_lut.h) instead of building the table; the
// driver only asks for them with kk <= 128 (ann_index.hip), so only those selections are instantiated.
template <int M, bool SLABBED, bool IMG = false, bool LAT = false>
static int32_t launch_scan_skew_m(const SkewArgs& sa, uint32_t n_blocks, uint64_t n_items, uint32_t kk, hipStream_t st) {
  auto lds_of = [&](int nw, int lr) { return sk_scan_lds(M, sa.res_floats, nw, lr); };
#define LAUNCH_SK_(LR, NT, MULTI, OPT, TWO, GRID)                                               \
  {                                                                                             \
    auto kern = k_scan_skew<M, LR, NT, MULTI, OPT, SLABBED, TWO, IMG, LAT>;                          \
    const size_t lds = lds_of(NT / 64, LR);                                                     \
    if (lds > 160u * 1024)                                                                      \
      return fail(MI355_ERR_NOT_SUPPORTED, "scan work item needs %zu B of LDS (> 160 KiB)", lds); \
    static std::atomic<int> table_at_lds_zero{0}; /* per instantiation: 0 unknown, 1 yes, -1 no */ \
    if (table_at_lds_zero.load(std::memory_order_relaxed) == 0) {                               \
      hipFuncAttributes fa;                                                                     \
      HIP_TRY(hipFuncGetAttributes(&fa, (const void*)kern));                                    \
      table_at_lds_zero.store(fa.sharedSizeBytes == 0 ? 1 : -1, std::memory_order_relaxed);     \
    }                                                                                           \
    if (table_at_lds_zero.load(std::memory_order_relaxed) < 0)                                  \
      return fail(MI355_ERR_NOT_SUPPORTED, "scan kernel was built with static LDS in front of its table " \
                  "(the gather address assumes the table at LDS address 0)");                   \
    HIP_TRY(ensure_dyn_lds((const void*)kern, lds));                                            \
    hipLaunchKernelGGL(kern, dim3(GRID), dim3(NT), lds, st, sa);                                \
  }
#define LAUNCH_SK(LR, NT, MULTI, OPT) LAUNCH_SK_(LR, NT, MULTI, OPT, false, n_blocks)
  if constexpr (LAT) {  // (the driver asks for these with k * refine_factor <= 64 only, ann_index.hip)
    // kk <= 16: lists of 128 rows (the waves' BEST rows bound the item); up to 128: lists of 192 and the waves' q-th bests
    if (kk > 128 || (kk > 16 && lds_of(16, 3) > 160u * 1024))
      return fail(MI355_ERR_NOT_SUPPORTED, "no sliced scan kernel for k * refine_factor = %u at this residual length", kk);
    if (kk <= 16) LAUNCH_SK(2, 1024, false, false)
    else LAUNCH_SK(3, 1024, false, false)
    HIP_TRY(hipGetLastError());
    return MI355_OK;
  } else {  // (else: a LAT launcher instantiates nothing below)
#ifndef SK_NO_PAIRED_WG
  // Tables of 32 columns (m <= 32: 128- to 512-d vectors at the reference's dim / 16) are 64 KiB, so TWO eight-wave
  // workgroups share a CU: a work item of so few columns spends a third of its time building its table and merging,
  // phases that leave the LDS gather and VALU pipes idle — the other workgroup's scan fills them.
  if constexpr (M == 32) {
    const uint32_t grid2 = (uint32_t)std::min<uint64_t>(2ull * n_blocks, std::max<uint64_t>(n_items, 1));
    if (kk <= 64 && 2 * lds_of(8, 2) <= 160u * 1024) {
      LAUNCH_SK_(2, 512, false, false, true, grid2)
      HIP_TRY(hipGetLastError());
      return MI355_OK;
    }
    if (kk > 64 && kk <= 128 && 2 * lds_of(8, 3) <= 160u * 1024) {  // (192-row lists: fits up to ~500 residual floats)
      LAUNCH_SK_(3, 512, false, false, true, grid2)
      HIP_TRY(hipGetLastError());
      return MI355_OK;
    }
  }
#endif
  // kk <= 128: sixteen waves with lists of 128 / 192 rows; beyond: sixteen waves with 192-row
  // lists and optimistic passes of SCAN_PASS_ROWS rows (k_scan_skew OPT) when that fits the LDS,
  // else eight waves with 320-row lists
  const bool opt_fits = lds_of(16, 3) <= 160u * 1024;
  if constexpr (IMG) {  // (no residual in LDS: the 16-wave lists always fit)
    if (kk <= 64) LAUNCH_SK(2, 1024, false, false)
    else if (kk <= 128) LAUNCH_SK(3, 1024, false, false)
    else return fail(MI355_ERR_NOT_SUPPORTED, "table images are scanned with k * refine_factor <= 128");
  } else {
  if (kk <= 64) LAUNCH_SK(2, 1024, false, false)
  else if (kk <= 128 && opt_fits) LAUNCH_SK(3, 1024, false, false)
  else if (kk <= 128) LAUNCH_SK(3, 512, false, false)  // a long residual: the lists of 16 waves do not fit
  else if (opt_fits) LAUNCH_SK(3, 1024, true, true)
  else if (kk <= SCAN_PASS_ROWS) LAUNCH_SK(5, 512, false, false)
  else LAUNCH_SK(5, 512, true, false)
  }
  HIP_TRY(hipGetLastError());
  return MI355_OK;
  }  // !LAT
#undef LAUNCH_SK
#undef LAUNCH_SK_
}
