// ann_scan_skew.hip — launcher of the production ADC scan kernel (k_scan_skew: pre-skewed code
// streams, conflict-free [code][column] table, persistent workgroups on per-XCD queues).
// Its own translation unit so that the kernel families compile in parallel.
#include "ann_scan_skew_impl.h"

int32_t launch_scan_skew_slab(const SkewArgs& sa, uint32_t M, uint32_t n_blocks, uint64_t n_items, uint32_t kk, hipStream_t st);

// M = columns per slab (SkewShape::M); slabbed: the padded / multi-slab kernel family (ann_scan_skew_slab.hip)
int32_t launch_scan_skew(const SkewArgs& sa, uint32_t M, uint32_t slabbed, uint32_t n_blocks, uint64_t n_items, uint32_t kk, hipStream_t st) {
  if (slabbed) return launch_scan_skew_slab(sa, M, n_blocks, n_items, kk, st);
  switch (M) {
    case 32: return launch_scan_skew_m<32, false>(sa, n_blocks, n_items, kk, st);
    case 48: return launch_scan_skew_m<48, false>(sa, n_blocks, n_items, kk, st);
    case 64: return launch_scan_skew_m<64, false>(sa, n_blocks, n_items, kk, st);
    case 80: return launch_scan_skew_m<80, false>(sa, n_blocks, n_items, kk, st);
    case 96: return launch_scan_skew_m<96, false>(sa, n_blocks, n_items, kk, st);
  }
  return fail(MI355_ERR_NOT_SUPPORTED, "no skewed scan kernel for a table of %u columns", M);
}
