// kernels_lut.h — batch-level PQ distance tables ("table images") for indexes whose sub-vectors are long.
//
// Why.  SURVEY.md §8a row a14: one distance table per (query, probed partition),
// LUT[j][c] = || (q - centroid_p)_j - codebook[j][c] ||^2.  Inside a scan work item (k_scan_skew, build_lut) every
// table streams the WHOLE codebook — 256 * dim * 4 bytes, 786 KB at 768-d whatever m is — out of L2 through one CU's
// 64 B/clk L1 fill: 12-19 us per item.  At the shape every stock LanceDB index has (partitions of ~8192 rows,
// rust/lancedb/src/table/create_index.rs:741-794; m = dim / 16, index/vector.rs:306-319) the item's code stream is
// 8192 * 48 B = 393 KB = 8 us of scan: the table build, not the scan, was the cost of a query.
//
// How.  The codebook is the operand that is REUSED by every pair of a batch, so it belongs in registers, and the
// pairs stream past it: a workgroup owns 16 table columns (sub-quantisers) for all 256 codes — 4 codebook entries of
// dsub floats per thread, loaded once — and loops over the batch's (query, partition) pairs.  A pair's residual slice
// is wave-uniform: it arrives in SGPRs (s_load from the residual array k_pair_residuals wrote) and costs no vector
// instruction.  The table leaves as an "image" [pair][slab][column block][code][16] f32 — 4 * m bytes per code row instead of 4 * dim
// bytes of codebook per row: 16 x less traffic per item at dsub = 16 — and the scan work item copies its image into the
// LDS table (kernels_skew.h load_lut_image: three 16-B loads per thread at M = 48).
//
// The arithmetic is the oracle's, entry by entry: r = q - c (one f32 subtraction), d = r - codebook, acc = fmaf(d, d, acc)
// over the dsub elements in order (dot: acc = fmaf(r, cb, acc), entry = 1 - acc); padding columns are 0.0f.  Same
// operations, same order, same bits as build_lut — tests/test_gpu_lut_images.py runs both paths against the oracle.
#pragma once
#include "kernels_ivfpq.h"

typedef __attribute__((ext_vector_type(4))) float lut_f32x4;
typedef __attribute__((ext_vector_type(2))) float lut_f32x2;

#define LUT_COLS_PER_WG 16u   // table columns a workgroup owns (one 64-byte run of an image row)
#define LUT_STAGE_PITCH 20u   // dwords per code row of the LDS staging tile (16 columns + padding: 16-B aligned quads)

// floats per row of the residual array: the columns in PAIRS, element-interleaved — [column pair][element t][column 2cp, 2cp + 1] —
// so that one s_load hands a wave the (r_2cp[t], r_2cp+1[t]) operand pairs of v_pk_add_f32 as they are (an odd m pads the
// last pair); behind them one word, the pair the row belongs to (+ padding to 64 B)
__host__ __device__ __forceinline__ uint32_t lut_res_floats(uint32_t m, uint32_t dsub) { return ((m + 1u) / 2u) * 2u * dsub; }
__host__ __device__ __forceinline__ uint32_t lut_res_stride(uint32_t m, uint32_t dsub) { return lut_res_floats(m, dsub) + 16u; }

// Row i of the residual array = the pair of the scan's i-th work item (of its first slice when pairs are sliced): the
// planner's list holds exactly the pairs that need a table — inactive query slots, probe ids outside the index, empty and
// foreign (sharded: 7 of 8) partitions make no item — and holds them partition-major, the order the scan consumes them in.
// res[i] = q_b - centroid_part (dot: q_b) in the layout above, read by every column block of k_lut_images through the
// scalar cache.  grid = an upper bound of the items (the pairs of the batch); the live count is q_start[8] / n_slices.
static __global__ __launch_bounds__(256) void k_pair_residuals(const float* __restrict__ qp, const float* __restrict__ centroids,
                                                               const SkewItem* __restrict__ items, const uint32_t* __restrict__ q_start,
                                                               uint32_t n_slices, uint32_t nprobe, uint32_t dim, uint32_t dsub,
                                                               uint32_t metric, float* __restrict__ res) {
  const uint32_t i = blockIdx.x;
  if (i >= q_start[8] / n_slices) return;
  const SkewItem it = items[(size_t)i * n_slices];
  const uint32_t pair = n_slices > 1u ? (it.pair & 0xFFFFFu) : it.pair;  // (sk_pack_pair)
  float* out = res + (size_t)i * lut_res_stride(dim / dsub, dsub);
  if (threadIdx.x == 0) ((uint32_t*)out)[lut_res_floats(dim / dsub, dsub)] = pair;
  const float* q = qp + (size_t)(pair / nprobe) * dim;
  const float* c = centroids + (size_t)it.part * dim;
  for (uint32_t d = threadIdx.x; d < dim; d += 256u) {
    const uint32_t j = d / dsub, t = d % dsub;
    out[(j >> 1) * 2u * dsub + 2u * t + (j & 1u)] = metric == MI355_METRIC_DOT ? q[d] : q[d] - c[d];
  }
}

// grid = (pair lanes, column blocks of 16, code halves): workgroup (y, x, z) builds columns 16x .. 16x + 15 (of the n_slabs * M
// image columns of a code row) of codes 128z .. 128z + 127 for the residual rows y, y + gridDim.x, ... of the batch.
//   8 waves; wave w: column pair w (columns 2w, 2w + 1 of the block), codes 128z + {lane, lane + 64}
//   -> 4 entries per thread, as two v_pk lanes (the two columns) x two codes; their dsub-float codebook vectors stay in
//   4 * DS VGPRs for the whole kernel; the residual of the column pair (2 * DS floats) is wave-uniform: SGPRs, fetched
//   one pair ahead (the s_load of pair i + 1 travels under the 16 * DS packed instructions of pair i).
// The 16 x 128 tile of a pair is staged through LDS so that it leaves as one contiguous 8 KiB run of the image (two staging
// tiles: one barrier per pair).  THREE such workgroups share a CU (<= 80 VGPRs): the phases of a pair — arithmetic, the
// barrier, the tile's read-back and store — are serial inside a workgroup, and a single 16-wave workgroup per CU left the
// VALU idle during the last two (2 GB of images at 3.1 TB/s where a plain fill writes 6.9).
#define LUT_NT 512
#define LUT_CODES_PER_WG 128u
template <int DS, bool DOT>
__global__ __launch_bounds__(LUT_NT, 6) void k_lut_images(const float* __restrict__ res, const float* __restrict__ codebook /*[m][256][DS]*/,
                                                          const uint32_t* __restrict__ q_start, uint32_t n_slices, uint32_t m, uint32_t M,
                                                          uint32_t n_slabs, uint32_t warm_ahead, uint32_t keep_mask, float* __restrict__ img) {
  __shared__ __attribute__((aligned(16))) float stage[2][LUT_CODES_PER_WG * LUT_STAGE_PITCH];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const uint32_t cp = w;
  const uint32_t col_blk = blockIdx.y, lane_y = blockIdx.x, n_lanes = gridDim.x;
  const uint32_t jg0 = col_blk * LUT_COLS_PER_WG;  // first image column of the block (slab * M + j)
  const uint32_t code0 = blockIdx.z * LUT_CODES_PER_WG;
  const uint32_t cl = lane;                           // this thread's codes inside the block: cl, cl + 64
  const uint32_t c0 = code0 + cl;
  // image column -> sub-quantiser of the index: slab s holds sub-quantisers s * M .. s * M + M - 1 (past m: padding)
  const uint32_t jcol = jg0 + 2u * cp;                // this wave's first column (uniform, even)
  const bool valid0 = jcol < m, valid1 = jcol + 1u < m;
  // the codebook vectors of the thread, the two columns side by side: cb[code i][t] = (cb[2cp][c][t], cb[2cp + 1][c][t])
  lut_f32x2 cb[2][DS];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint32_t jq0 = valid0 ? jcol : 0u, jq1 = valid1 ? jcol + 1u : 0u;  // (a padding column reads column 0: never used)
    const lut_f32x4* s0 = (const lut_f32x4*)(codebook + ((size_t)jq0 * 256u + c0 + 64u * i) * DS);
    const lut_f32x4* s1 = (const lut_f32x4*)(codebook + ((size_t)jq1 * 256u + c0 + 64u * i) * DS);
#pragma unroll
    for (int v = 0; v < DS / 4; ++v) {
      const lut_f32x4 x = s0[v], y = s1[v];
      cb[i][4 * v + 0] = lut_f32x2{x.x, y.x};
      cb[i][4 * v + 1] = lut_f32x2{x.y, y.y};
      cb[i][4 * v + 2] = lut_f32x2{x.z, y.z};
      cb[i][4 * v + 3] = lut_f32x2{x.w, y.w};
    }
  }
  const uint32_t rs = lut_res_stride(m, DS);
  const uint32_t roff = (valid0 ? jcol >> 1 : 0u) * 2u * DS;  // this wave's column pair inside a residual row
  const uint32_t slab = jg0 / M, j0 = jg0 % M;              // (16 | M: a block never straddles two slabs)
  const uint32_t pair_at = lut_res_floats(m, DS);
  const uint32_t n_rows = q_start[8] / n_slices;  // live pairs of the batch (uniform)
  auto load_r = [&](uint32_t row_i, lut_f32x2 (&r)[DS], uint32_t& pair) {  // uniform address, read-only array: scalar loads
    const float* row = res + (size_t)row_i * rs;
    const lut_f32x2* src = (const lut_f32x2*)(row + roff);
#pragma unroll
    for (int t = 0; t < DS; ++t) r[t] = src[t];
    pair = ((const uint32_t*)row)[pair_at];
  };
  const size_t img_pair_stride = (size_t)n_slabs * 256u * M;
  const uint32_t tile_off = slab * 256u * M + (j0 / LUT_COLS_PER_WG) * (256u * LUT_COLS_PER_WG) + code0 * LUT_COLS_PER_WG;
  uint32_t buf = 0;
  auto one_pair = [&](const lut_f32x2 (&r)[DS], uint32_t pair) {
    lut_f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < DS; ++t) {
      if (DOT) {
        a0 = __builtin_elementwise_fma(r[t], cb[0][t], a0);
        a1 = __builtin_elementwise_fma(r[t], cb[1][t], a1);
      } else {
        const lut_f32x2 d0 = r[t] - cb[0][t], d1 = r[t] - cb[1][t];
        a0 = __builtin_elementwise_fma(d0, d0, a0);
        a1 = __builtin_elementwise_fma(d1, d1, a1);
      }
    }
    if (DOT) {
      a0 = 1.0f - a0;
      a1 = 1.0f - a1;
    }
    if (!valid0) a0.x = a1.x = 0.f;  // a padding column: `+ 0.0f` is exact in the row sum
    if (!valid1) a0.y = a1.y = 0.f;
    float* st = stage[buf];
    *(lut_f32x2*)(st + cl * LUT_STAGE_PITCH + 2u * cp) = a0;
    *(lut_f32x2*)(st + (cl + 64u) * LUT_STAGE_PITCH + 2u * cp) = a1;
    __syncthreads();
    // quad `tid` of the tile: code tid / 4, columns 4 * (tid % 4) .. + 3
    const uint32_t c2 = tid >> 2, q2 = tid & 3u;
    const lut_f32x4 v = *(const lut_f32x4*)(st + c2 * LUT_STAGE_PITCH + 4u * q2);
    // image layout [pair][slab][column block][code][16 columns]: the tile of this workgroup is ONE contiguous 8 KiB run
    // (thread e stores bytes 16 e ..), not 128 64-byte pieces 4 * M bytes apart
    // (a uniform base + a 32-bit thread offset: the store takes its base from SGPRs, no 64-bit address per thread)
    float* tile = img + (size_t)((keep_mask & 4u) ? 0u : pair) * img_pair_stride + tile_off;  // (bit 2: dev, every pair stores over pair 0's image)
#ifndef LUT_NT_STORE
#define LUT_NT_STORE 1  // (streaming stores: the images are read once, by another kernel; -50 us per 2 GB of images)
#endif
    if (!(keep_mask & 2u)) {  // (bit 1: dev, no stores)
      if (LUT_NT_STORE) __builtin_nontemporal_store(v, (lut_f32x4*)(tile + 4u * tid));
      else *(lut_f32x4*)(tile + 4u * tid) = v;
    }
    buf ^= 1u;  // (the other tile: every thread read its quad of THIS tile before the next barrier)
  };
  // The residual rows of a batch (3 KB each, 126 MB at 2048 x 20 pairs) sit in HBM / the Infinity Cache when this kernel
  // starts, and a scalar load that misses L2 takes longer than the one pair it is issued ahead.  So the workgroup warms L2:
  // `warm_ahead` pairs ahead, 16 lanes of wave 0 touch the 128-byte lines of the block's slice of two rows with ordinary
  // vector loads.  A touched word is consumed one whole loop iteration (four pairs) later — OR-ed into `wacc` under
  // `keep_mask`, a kernel argument that is always 0: the compiler must keep the loads and tracks them with counted vmcnt
  // waits, and nothing is ever stored.  (An asm statement or an LDS-DMA intrinsic would wait for nothing at all, but with
  // either in the kernel the compiler can no longer prove that the residual array is not written and turns the SCALAR
  // loads above into vector loads.)
  const bool toucher = warm_ahead != 0u && w == 0u && lane < 16u;  // lanes 0-7: the first row of a pair of rows, 8-15: the second
  const uint32_t blk_off = (jg0 < m ? jg0 >> 1 : 0u) * 2u * DS;   // the block's 16 columns inside a residual row (1 KB)
  const uint32_t step = n_lanes;
  // the block's rows: blockIdx.y + k * step, k < n_mine; every loop iteration runs four of them, clamped to the last one
  // (a tail builds and stores its last row again, the same bytes: no conditional code the scalar loads could sink into)
  const uint32_t n_mine = n_rows > lane_y ? (n_rows - lane_y + step - 1u) / step : 0u;
  auto row_of = [&](uint32_t k) { return lane_y + (k < n_mine ? k : n_mine - 1u) * step; };
  auto touch = [&](uint32_t k) -> float {
    float t = 0.f;
    if (toucher) t = res[(size_t)row_of(k + warm_ahead + (lane >> 3)) * rs + blk_off + (lane & 7u) * 32u];
    return t;
  };
  // two residual register sets, ping-pong: no copies between them
  lut_f32x2 ra[DS], rb[DS];
  uint32_t pa, pb, wacc = 0;
  float t0 = 0.f, t1 = 0.f;
  if (n_mine) load_r(row_of(0), ra, pa);
  auto two_pairs = [&](uint32_t k) {
    load_r(row_of(k + 1u), rb, pb);
    __builtin_amdgcn_sched_barrier(0);  // (the scheduler would sink the loads behind the arithmetic to save SGPRs ...
    one_pair(ra, pa);
    __builtin_amdgcn_sched_barrier(0);  //  ... and hoist the next ones above the LDS wait of the tile's read-back)
    load_r(row_of(k + 2u), ra, pa);
    __builtin_amdgcn_sched_barrier(0);
    one_pair(rb, pb);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (uint32_t k = 0; k < n_mine; k += 4u) {  // workgroup-uniform
    wacc |= __float_as_uint(t0) & (keep_mask & 1u);
    t0 = touch(k);
    two_pairs(k);
    wacc |= __float_as_uint(t1) & (keep_mask & 1u);
    t1 = touch(k + 2u);
    two_pairs(k + 2u);
  }
  wacc |= (__float_as_uint(t0) | __float_as_uint(t1)) & (keep_mask & 1u);
  if (wacc) img[0] = 0.f;  // (never: keep_mask is 0)
}
