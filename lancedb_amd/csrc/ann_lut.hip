// ann_lut.hip — launcher of the batch-level distance-table kernels (kernels_lut.h: k_pair_residuals, k_lut_images).
// SURVEY.md section 8a row a14 (one PQ distance table per (query, probed partition)) for the index shape the reference
// builds by default: m = dim / 16 (rust/lancedb/src/index/vector.rs:306-319), partitions of ~8192 rows
// (table/create_index.rs:741-794), where a work item's table build used to cost more than its scan.
#include "ann_internal.h"
#include "kernels_ivfpq.h"
#include "kernels_skew.h"
#include "kernels_lut.h"

// 8-bit codes on the production scan with sub-vectors of 16 floats (the kernels are instantiated for that length)
bool lut_images_shape_ok(const mi355_index* ix) {
  const bool ds_ok = ix->dsub == 16 || (ix->dsub == 8 && dev_knob("MI355_LUT_IMAGES_DS8", 0));  // (dsub 8: dev A/B, see NOTES 11.8)
  return ix->layout == MI355_SCAN_SKEW && ix->nbits == 8 && ds_ok && ix->m * ix->dsub == ix->dim;
}
size_t lut_image_bytes_per_pair(const mi355_index* ix) { return (size_t)ix->sk_slabs * 256u * ix->sk_M * sizeof(float); }
size_t lut_residual_bytes_per_pair(const mi355_index* ix) { return (size_t)lut_res_stride(ix->m, ix->dsub) * sizeof(float); }

// the tables of the work list `items` (q_start[8] items, n_slices per pair) -> img [pair][slab][256][M]
int32_t launch_lut_images(mi355_index* ix, const float* qp, const SkewItem* items, const uint32_t* q_start, uint32_t n_pairs,
                          uint32_t n_slices, uint32_t nprobe, float* res, float* img, hipStream_t st) {
  if (!n_pairs) return MI355_OK;
  hipLaunchKernelGGL(k_pair_residuals, dim3(n_pairs), dim3(256), 0, st, qp, ix->centroids.as<float>(), items, q_start, n_slices, nprobe,
                     ix->dim, ix->dsub, ix->metric, res);
  // one 16-wave workgroup per CU holds its codebook slice in registers and walks the residual rows
  const uint32_t col_blocks = ix->sk_slabs * ix->sk_M / LUT_COLS_PER_WG, halves = 256u / LUT_CODES_PER_WG;
  // three 8-wave workgroups per CU, each with its codebook slice in registers, walk the residual rows
  uint32_t lanes = std::max(1u, 3u * ix->n_cus / (col_blocks * halves));
  lanes = std::max(1u, std::min(n_pairs, lanes * dev_knob("MI355_LUT_LANES_X", 1)));
  const dim3 grid(lanes, col_blocks, halves);
  const uint32_t warm = dev_knob("MI355_LUT_WARM_AHEAD", 8);  // (pairs ahead of the L2 warm-up, 0 = off)
  const uint32_t dbg = dev_knob("MI355_LUT_DBG", 0) << 1;      // (dev: 1 = no image stores, 2 = all stores over one image)
#define LAUNCH_LUT(DS, DOT)                                                                                                            \
  hipLaunchKernelGGL((k_lut_images<DS, DOT>), grid, dim3(LUT_NT), 0, st, res, ix->codebook.as<float>(), q_start, n_slices, ix->m, ix->sk_M, \
                     ix->sk_slabs, warm, dbg, img)
  const bool dot = ix->metric == MI355_METRIC_DOT;
  if (ix->dsub == 16) {
    if (dot) LAUNCH_LUT(16, true); else LAUNCH_LUT(16, false);
  } else {
    if (dot) LAUNCH_LUT(8, true); else LAUNCH_LUT(8, false);
  }
#undef LAUNCH_LUT
  HIP_TRY(hipGetLastError());
  return MI355_OK;
}
