// kernels_flat.h — exact (no index / bypass_vector_index) KNN on a raw column.
// Replaces KNNVectorDistance + SortExec TopK
// (/root/reference/python/python/lancedb/query.py:1365-1370).
#pragma once
#include "kernels_ivfpq.h"

struct FlatArgs {
  const void* vectors;      // [n_rows, dim]
  uint32_t dtype;
  const uint64_t* row_ids;  // or nullptr
  uint64_t n_rows;
  uint32_t dim, metric;
  const float* q;           // [nq, dim] original queries
  uint32_t slice_rows;      // rows per work item
  uint32_t n_slices;
  uint32_t kk;
  RangeFilter range;
  RowFilter filter;
  Cand* cand;               // [nq, n_slices, kk]
};

// One workgroup per (slice, query).  Every row's distance is the sequential
// d-ascending chain of the contract, one row per thread; a wave therefore
// touches 64 rows x 16 B per load and walks each row's cache lines in order.
template <int KPL>
__global__ __launch_bounds__(256) void k_flat_scan(FlatArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sq = (float*)smem;  // [dim]
  Cand* stage = (Cand*)(smem + (((size_t)a.dim * 4 + 15) & ~(size_t)15));  // [4][kk]
  __shared__ float s_qq;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t s = blockIdx.x, b = blockIdx.z;
  const float* q = a.q + (size_t)b * a.dim;
  for (uint32_t d = tid; d < a.dim; d += 256) sq[d] = q[d];
  __syncthreads();
  if (tid == 0) {
    float acc = 0.f;
    for (uint32_t d = 0; d < a.dim; ++d) acc = __fmaf_rn(sq[d], sq[d], acc);
    s_qq = acc;
  }
  __syncthreads();
  const float qq = s_qq;
  const uint64_t v0 = (uint64_t)s * a.slice_rows;
  const uint64_t v1 = min(a.n_rows, v0 + (uint64_t)a.slice_rows);
  WaveTopK<KPL> top;
  top.init(a.kk, lane);
  for (uint64_t i0 = v0; i0 < v1; i0 += 256) {
    uint64_t i = i0 + tid;
    bool ok = i < v1;
    float d = 0.f;
    if (ok) {
      d = exact_distance(sq, a.vectors, a.dtype, i, a.dim, a.metric, qq);
      ok = d <= top.thr_d && in_range(d, a.range);
    }
    if (__any(ok)) {
      uint64_t id = 0;
      if (ok) id = a.row_ids ? a.row_ids[i] : i;
      if (a.filter.mode != MI355_FILTER_NONE && ok) ok = row_permitted(id, a.filter);
      top.offer(ok, d, (uint32_t)i, id, lane);
    }
  }
  top.store(stage + (size_t)wid * a.kk, lane);
  __syncthreads();
  if (wid == 0) {
    const uint32_t n = 3 * a.kk;
    for (uint32_t t0 = 0; t0 < n; t0 += MI355_WAVE) {
      uint32_t t = t0 + lane;
      Cand c;
      c.d = 0.f;
      c.pos = CAND_EMPTY_POS;
      c.id = 0;
      if (t < n) c = stage[a.kk + t];
      top.offer(t < n && c.pos != CAND_EMPTY_POS, c.d, c.pos, c.id, lane);
    }
    top.store(a.cand + ((size_t)b * a.n_slices + s) * a.kk, lane);
  }
}
