// kernels_encode.h — index population on the GPU (mi355_ivfpq_encode): partition
// assignment, residual PQ encoding and the stable partition order of every row.
// This is the O(N) transform stage of the reference's index build
// (/root/reference/rust/lancedb/src/table/create_index.rs:114-151, :283-303: the
// builder hands IvfBuildParams + PQBuildParams to lance, which trains on samples and
// then transforms + shuffles ALL rows [EXT]); restated in oracle/ann_oracle.c
// (orc_ivfpq_encode), bit-exact parity in tests/test_gpu_encode.py.
//
//   pass A  k_prep_queries + k_coarse_mfma (the search's own coarse stage, f32 MFMA =
//           the contract's fmaf chain) over chunks of rows, k_argmin_rows -> partition
//           of every row + histogram
//   pass B  stable counting sort without a global sort: per 256-row block the rank of a
//           row among the earlier rows of its partition (k_local_rank), a dense
//           [block][partition] count matrix scanned per partition over the blocks of a
//           chunk (k_block_scan, carrying the running offsets from chunk to chunk)
//           and, in the same chunk (the rows cross PCIe / HBM once), k_encode_rows: one
//           thread per (row, 4 sub-quantisers), the codebook slices in LDS (every lane
//           reads the same entry: broadcast, conflict-free), the contract's chain_l2
//           against all 256 entries; codes land in SOURCE order
//   pass C  k_permute_codes: code rows gathered into index order
#pragma once
#include "kernels_ivfpq.h"

// ---- pass A: arg-min over the coarse distances of one row --------------------------
// NaN never wins; among equal minima the lowest partition id wins (oracle: first minimum)
static __global__ __launch_bounds__(256) void k_argmin_rows(const float* __restrict__ coarse, uint32_t n, uint32_t nlist,
                                                     uint32_t* __restrict__ assign, uint32_t* __restrict__ hist) {
  __shared__ uint32_t s_key[4], s_idx[4];
  const uint32_t row = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* src = coarse + (size_t)row * nlist;
  uint32_t bk = 0xFFFFFFFFu, bi = 0xFFFFFFFFu;  // sort key (NaN = 0xFFFFFFFF loses), index
  for (uint32_t p = tid; p < nlist; p += 256) {
    const float v = src[p];
    const uint32_t k = f32_sort_key(v);
    if (v == v && (k < bk || (k == bk && p < bi))) {
      bk = k;
      bi = p;
    }
  }
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t ok = __shfl_xor(bk, off), oi = __shfl_xor(bi, off);
    if (ok < bk || (ok == bk && oi < bi)) {
      bk = ok;
      bi = oi;
    }
  }
  if (lane == 0) {
    s_key[wid] = bk;
    s_idx[wid] = bi;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (s_key[w] < bk || (s_key[w] == bk && s_idx[w] < bi)) {
        bk = s_key[w];
        bi = s_idx[w];
      }
    if (bi == 0xFFFFFFFFu) bi = 0;  // every distance NaN: partition 0, as the oracle
    assign[row] = bi;
    atomicAdd(&hist[bi], 1u);
  }
}

// ---- pass B: stable positions ---------------------------------------------------------
// block b of a chunk covers rows [r0 + 256 b, +256): rank of each row among the earlier
// rows of the same partition in the block; the first row of a partition records the count
static __global__ __launch_bounds__(256) void k_local_rank(const uint32_t* __restrict__ assign, uint64_t r0, uint64_t n_rows,
                                                    uint32_t nlist, uint32_t* __restrict__ cntB,
                                                    uint32_t* __restrict__ lrank) {
  __shared__ uint32_t s_a[256];
  const int tid = threadIdx.x;
  const uint64_t row = r0 + (uint64_t)blockIdx.x * 256 + tid;
  const uint32_t mine = row < n_rows ? assign[row] : 0xFFFFFFFFu;
  s_a[tid] = mine;
  __syncthreads();
  if (mine == 0xFFFFFFFFu) return;
  uint32_t before = 0, total = 0;
  for (int j = 0; j < 256; ++j) {
    const bool same = s_a[j] == mine;
    total += same ? 1u : 0u;
    before += (same && j < tid) ? 1u : 0u;
  }
  lrank[(size_t)blockIdx.x * 256 + tid] = before;
  if (before == 0) cntB[(size_t)blockIdx.x * nlist + mine] = total;
}

// one thread per partition: exclusive scan of its counts over the chunk's blocks; the
// partition's running offset advances into run_out (k_positions still needs run_in)
static __global__ void k_block_scan(uint32_t* __restrict__ cntB, uint32_t n_blocks, uint32_t nlist,
                             const unsigned long long* __restrict__ run_in,
                             unsigned long long* __restrict__ run_out) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nlist) return;
  // offsets relative to the partition's running offset fit 32 bits (a chunk has <= 65536 rows)
  uint32_t rel = 0;
  for (uint32_t b = 0; b < n_blocks; ++b) {
    const uint32_t t = cntB[(size_t)b * nlist + p];
    cntB[(size_t)b * nlist + p] = rel;
    rel += t;
  }
  run_out[p] = run_in[p] + rel;
}

static __global__ __launch_bounds__(256) void k_positions(const uint32_t* __restrict__ assign, uint64_t r0, uint64_t n_rows,
                                                   uint32_t nlist, const uint32_t* __restrict__ cntB,
                                                   const uint32_t* __restrict__ lrank,
                                                   const unsigned long long* __restrict__ chunk_base,
                                                   uint64_t* __restrict__ order) {
  const uint64_t row = r0 + (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= n_rows) return;
  const uint32_t p = assign[row];
  const uint64_t at = chunk_base[p] + cntB[(size_t)blockIdx.x * nlist + p] + lrank[(size_t)blockIdx.x * 256 + threadIdx.x];
  order[at] = row;
}

// ---- residual PQ encoding ------------------------------------------------------------
struct EncodeArgs {
  const float* x;          // [n, dim] rows of this chunk, already normalised for cosine (k_prep_queries)
  uint64_t r0, n;          // first source row and row count of this chunk
  const uint32_t* assign;  // [n_total]
  const float* centroids;  // [nlist, dim]
  const float* codebook;   // [m, 2^NBITS, dsub]
  uint32_t dim, m, dsub, metric;
  uint32_t mb;             // code bytes per row: m * NBITS / 8
  uint8_t* codes;          // [n_total, mb] SOURCE order
};

// grid (row blocks of 256, m / JT): thread = (row, JT consecutive sub-quantisers).
// NBITS = 4 (JT = 2): two codes per byte, sub-quantiser 2t in the low nibble of byte t.
template <int JT, int NBITS>
__global__ __launch_bounds__(256) void k_encode_rows(EncodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr uint32_t KS = 1u << NBITS;
  static_assert(NBITS == 8 || (NBITS == 4 && JT == 2), "4-bit codes are encoded a byte (two sub-quantisers) at a time");
  float* cb = (float*)smem;  // [JT][KS][dsub]
  const int tid = threadIdx.x;
  const uint32_t j0 = blockIdx.y * JT, dsub = a.dsub;
  for (uint32_t e = tid; e < (uint32_t)JT * KS * dsub; e += 256) cb[e] = a.codebook[(size_t)j0 * KS * dsub + e];
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * 256 + tid;
  if (i >= a.n) return;
  const uint64_t row = a.r0 + i;
  const uint32_t p = a.assign[row];
  const float* xs = a.x + (size_t)i * a.dim + (size_t)j0 * dsub;
  const float* cs = a.centroids + (size_t)p * a.dim + (size_t)j0 * dsub;
  const bool dotm = a.metric == MI355_METRIC_DOT;
  uint32_t packed = 0;
  for (int jj = 0; jj < JT; ++jj) {
    // residual of this sub-vector: registers for the common sizes, re-derived per entry otherwise
    float r[16];
    const bool small = dsub <= 16;
    if (small)
#pragma unroll
      for (uint32_t t = 0; t < 16; ++t)
        if (t < dsub) r[t] = dotm ? xs[jj * dsub + t] : xs[jj * dsub + t] - cs[jj * dsub + t];
    uint32_t bc = 0;
    float bv = 0.f;
    bool have = false;
    const float* cj = cb + (size_t)jj * KS * dsub;
    for (uint32_t c = 0; c < KS; ++c) {
      float acc = 0.f;
      auto step = [&](float rv, float e) {  // e: same LDS address in every lane (broadcast)
        if (dotm)
          acc = __fmaf_rn(rv, e, acc);
        else {
          const float df = rv - e;
          acc = __fmaf_rn(df, df, acc);
        }
      };
      if (small) {
#pragma unroll
        for (uint32_t t = 0; t < 16; ++t)
          if (t < dsub) step(r[t], cj[c * dsub + t]);
      } else {
        for (uint32_t t = 0; t < dsub; ++t)
          step(dotm ? xs[jj * dsub + t] : xs[jj * dsub + t] - cs[jj * dsub + t], cj[c * dsub + t]);
      }
      const float v = dotm ? 1.0f - acc : acc;
      if (v == v && (!have || v < bv)) {
        bv = v;
        bc = c;
        have = true;
      }
    }
    packed |= bc << (NBITS * jj);
  }
  if (NBITS == 4) {
    a.codes[(size_t)row * a.mb + j0 / 2] = (uint8_t)packed;
    return;
  }
  uint8_t* dst = a.codes + (size_t)row * a.mb + j0;
  if (JT == 4)
    *(uint32_t*)dst = packed;  // m % 4 == 0: the row's codes are 4-byte aligned
  else
    dst[0] = (uint8_t)packed;
}

// ---- pass C: code rows into index order ------------------------------------------------
// one thread per 4 code bytes when m % 4 == 0 (W = 4), per byte otherwise
template <int W>
__global__ void k_permute_codes(const uint8_t* __restrict__ src, const uint64_t* __restrict__ order, uint64_t n,
                                uint32_t m, uint8_t* __restrict__ dst) {
  const uint32_t per = m / W;
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n * per) return;
  const uint64_t at = g / per;
  const uint32_t w = (uint32_t)(g - at * per);
  const uint64_t row = order[at];
  if (W == 4)
    ((uint32_t*)dst)[at * per + w] = ((const uint32_t*)src)[row * per + w];
  else
    dst[at * per + w] = src[row * per + w];
}

// ---- index training (mi355_kmeans_train / mi355_ivf_residuals) -------------------------
// Lloyd update: one thread per (centroid, dimension) adds the centroid's rows ONE BY ONE in
// source-row order (the stable order of pass B) — a sum whose order is part of the
// definition, so the trained centroids are bit-exact against the oracle; loads run four
// rows ahead of the adds.
static __global__ __launch_bounds__(256) void k_centroid_update(const float* __restrict__ xp,
                                                         const uint64_t* __restrict__ order,
                                                         const unsigned long long* __restrict__ po, uint32_t dim,
                                                         float* __restrict__ cen) {
  const uint32_t p = blockIdx.x;
  const uint32_t d = blockIdx.y * 256 + threadIdx.x;
  const unsigned long long lo = po[p], hi = po[p + 1];
  if (d >= dim || hi == lo) return;  // a centroid without rows keeps its value
  float acc = 0.f;
  unsigned long long i = lo;
  for (; i + 4 <= hi; i += 4) {
    const float v0 = xp[(size_t)order[i] * dim + d], v1 = xp[(size_t)order[i + 1] * dim + d];
    const float v2 = xp[(size_t)order[i + 2] * dim + d], v3 = xp[(size_t)order[i + 3] * dim + d];
    acc = acc + v0;
    acc = acc + v1;
    acc = acc + v2;
    acc = acc + v3;
  }
  for (; i < hi; ++i) acc = acc + xp[(size_t)order[i] * dim + d];
  cen[(size_t)p * dim + d] = ieee_divf(acc, (float)(hi - lo));
}

static __global__ void k_residuals(const float* __restrict__ xp, const uint32_t* __restrict__ assign,
                            const float* __restrict__ cen, uint64_t n, uint32_t dim, uint32_t metric,
                            float* __restrict__ out) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n * dim) return;
  const uint64_t row = g / dim;
  const uint32_t d = (uint32_t)(g - row * dim);
  const float v = xp[g];
  out[g] = metric == MI355_METRIC_DOT ? v : v - cen[(size_t)assign[row] * dim + d];
}
