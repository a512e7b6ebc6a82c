// ann_scan_skew_slab.hip — the padded / multi-slab instantiations of k_scan_skew (SkewShape, kernels_skew.h): every
// 8-bit num_sub_vectors the reference's builder produces (rust/lancedb/src/index/vector.rs:306-319: dim / 16, dim / 8
// or 1 — 24 for 384-d, 60 for 960-d, 128 for 2048-d, 192 for 3072-d) runs the production scan, not the generic one.
#include "ann_scan_skew_impl.h"

int32_t launch_scan_skew_slab(const SkewArgs& sa, uint32_t M, uint32_t n_blocks, uint64_t n_items, uint32_t kk, hipStream_t st) {
  switch (M) {
    case 32: return launch_scan_skew_m<32, true>(sa, n_blocks, n_items, kk, st);
    case 48: return launch_scan_skew_m<48, true>(sa, n_blocks, n_items, kk, st);
    case 64: return launch_scan_skew_m<64, true>(sa, n_blocks, n_items, kk, st);
    case 80: return launch_scan_skew_m<80, true>(sa, n_blocks, n_items, kk, st);
    case 96: return launch_scan_skew_m<96, true>(sa, n_blocks, n_items, kk, st);
  }
  return fail(MI355_ERR_NOT_SUPPORTED, "no skewed scan kernel for a table of %u columns", M);
}
