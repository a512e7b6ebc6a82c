// ann_scan_skew_slab_img.hip — the table-image instantiations of k_scan_skew for the padded / multi-slab widths
// (m = dim / 16 of 384-d, 960-d, 2048-d, 3072-d vectors: rust/lancedb/src/index/vector.rs:306-319).
#include "ann_scan_skew_impl.h"

int32_t launch_scan_skew_slab_img(const SkewArgs& sa, uint32_t M, uint32_t n_blocks, uint64_t n_items, uint32_t kk, hipStream_t st) {
  switch (M) {
    case 32: return launch_scan_skew_m<32, true, true>(sa, n_blocks, n_items, kk, st);
    case 48: return launch_scan_skew_m<48, true, true>(sa, n_blocks, n_items, kk, st);
    case 64: return launch_scan_skew_m<64, true, true>(sa, n_blocks, n_items, kk, st);
    case 80: return launch_scan_skew_m<80, true, true>(sa, n_blocks, n_items, kk, st);
    case 96: return launch_scan_skew_m<96, true, true>(sa, n_blocks, n_items, kk, st);
  }
  return fail(MI355_ERR_NOT_SUPPORTED, "no skewed scan kernel for a table of %u columns", M);
}
