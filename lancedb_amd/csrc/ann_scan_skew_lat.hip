// ann_scan_skew_lat.hip — the LAT instantiations of k_scan_skew (kernels_skew.h): what a sliced batch runs — a few queries cut into
// about one work item per CU, every item starting without a bound.  Its own translation unit: compiles in parallel.
#include "ann_scan_skew_impl.h"

int32_t launch_scan_skew_lat(const SkewArgs& sa, uint32_t M, uint32_t n_blocks, uint64_t n_items, uint32_t kk, hipStream_t st) {
  switch (M) {
    case 32: return launch_scan_skew_m<32, false, false, true>(sa, n_blocks, n_items, kk, st);
    case 48: return launch_scan_skew_m<48, false, false, true>(sa, n_blocks, n_items, kk, st);
    case 64: return launch_scan_skew_m<64, false, false, true>(sa, n_blocks, n_items, kk, st);
    case 80: return launch_scan_skew_m<80, false, false, true>(sa, n_blocks, n_items, kk, st);
    case 96: return launch_scan_skew_m<96, false, false, true>(sa, n_blocks, n_items, kk, st);
  }
  return fail(MI355_ERR_NOT_SUPPORTED, "no skewed scan kernel for a table of %u columns", M);
}
