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
// kk > 64 * KPL: the slice is swept once per pass of 64 * KPL rows, each pass keeping the
// best rows strictly above the worst row of the previous one (WaveTopK floor).
template <int KPL>
__global__ __launch_bounds__(256) void k_flat_scan(FlatArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sq = (float*)smem;  // [dim]
  Cand* stage = (Cand*)(smem + (((size_t)a.dim * 4 + 15) & ~(size_t)15));  // [4][min(kk, 64 KPL)]
  __shared__ float s_qq;
  __shared__ PassFloor s_floor;
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
  Cand* out = a.cand + ((size_t)b * a.n_slices + s) * a.kk;
  constexpr uint32_t C = KPL * MI355_WAVE;
  bool fl_on = false;
  float fl_d = 0.f;
  uint64_t fl_id = 0;
  for (uint32_t base = 0; base < a.kk; base += C) {
    const uint32_t c = min(a.kk - base, C);
    WaveTopK<KPL> top;
    top.init(c, lane);
    top.set_floor(fl_on, fl_d, (uint32_t)fl_id, (uint32_t)(fl_id >> 32));
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
    top.store(stage + (size_t)wid * c, lane);
    __syncthreads();
    if (wid == 0) {
      const uint32_t n = 3 * c;
      for (uint32_t t0 = 0; t0 < n; t0 += MI355_WAVE) {
        uint32_t t = t0 + lane;
        Cand cd;
        cd.d = 0.f;
        cd.pos = CAND_EMPTY_POS;
        cd.id = 0;
        if (t < n) cd = stage[c + t];
        top.offer(t < n && cd.pos != CAND_EMPTY_POS, cd.d, cd.pos, cd.id, lane);
      }
      top.store(out + base, lane);
      // the slice's pass is full iff its worst kept slot is a real row: the next pass starts above it
      const bool full = !(top.thr_d == __builtin_huge_valf() && top.thr_lo == 0xFFFFFFFFu && top.thr_hi == 0xFFFFFFFFu);
      if (lane == 0) {
        s_floor.on = full ? 1u : 0u;
        s_floor.d = top.thr_d;
        s_floor.id = ((uint64_t)top.thr_hi << 32) | top.thr_lo;
      }
    }
    __syncthreads();
    if (!s_floor.on) {  // fewer rows than asked for: the remaining slots are empty
      for (uint32_t g = base + c + tid; g < a.kk; g += 256) {
        Cand e;
        e.d = __builtin_huge_valf();
        e.pos = CAND_EMPTY_POS;
        e.id = ~0ull;
        out[g] = e;
      }
      break;
    }
    fl_on = true;
    fl_d = s_floor.d;
    fl_id = s_floor.id;
    __syncthreads();  // stage is rewritten by the next pass
  }
}
