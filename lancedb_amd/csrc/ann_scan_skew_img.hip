// ann_scan_skew_img.hip — the table-image instantiations of k_scan_skew (IMG: the work item copies the distance table
// k_lut_images built for its pair, kernels_lut.h) for the plain widths.  Its own translation unit: compiles in parallel.
#include "ann_scan_skew_impl.h"

int32_t launch_scan_skew_slab_img(const SkewArgs& sa, uint32_t M, uint32_t n_blocks, uint64_t n_items, uint32_t kk, hipStream_t st);

int32_t launch_scan_skew_img(const SkewArgs& sa, uint32_t M, uint32_t slabbed, uint32_t n_blocks, uint64_t n_items, uint32_t kk, hipStream_t st) {
  if (slabbed) return launch_scan_skew_slab_img(sa, M, n_blocks, n_items, kk, st);
  switch (M) {
    case 32: return launch_scan_skew_m<32, false, true>(sa, n_blocks, n_items, kk, st);
    case 48: return launch_scan_skew_m<48, false, true>(sa, n_blocks, n_items, kk, st);
    case 64: return launch_scan_skew_m<64, false, true>(sa, n_blocks, n_items, kk, st);
    case 80: return launch_scan_skew_m<80, false, true>(sa, n_blocks, n_items, kk, st);
    case 96: return launch_scan_skew_m<96, false, true>(sa, n_blocks, n_items, kk, st);
  }
  return fail(MI355_ERR_NOT_SUPPORTED, "no skewed scan kernel for a table of %u columns", M);
}
