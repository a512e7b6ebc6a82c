// kernels_ivfpq.h — gfx950 kernels of the IVF-PQ search path.
//
//   K0 prep_queries     cosine normalisation + |q|^2            (per query)
//   K1 coarse_tile      q x centroid fma chains -> coarse dist  (E2, table/query.rs:1079)
//   K1b select_probes   radix-select the nprobe nearest partitions
//   K2+K3+K4 scan_pair  LUT build in LDS + ADC scan + wave top-k (E3/E5/E6)
//   K4 merge_cands      per-query reducer over all scan work items (E8)
//   K6 refine           exact re-rank on raw vectors             (E9)
//
// Reference behaviour restated in oracle/ann_oracle.c; every kernel is parity
// tested against it bit for bit (tests/test_gpu_parity.py).
#pragma once
#include "device_common.h"

// ---------------------------------------------------------------------------
// device view of an opened index
// ---------------------------------------------------------------------------
struct IndexView {
  uint32_t dim, nlist, m, dsub, metric;
  uint32_t nbits, mb;         // PQ bits per code (8 | 4) and code bytes per row (m * nbits / 8)
  const float* centroids;     // [nlist, dim]
  const float* cnorm;         // [nlist] chain_dot(c,c)
  const float* codebook;      // [m, 2^nbits, dsub]
  const uint8_t* codes;       // partition p: [mb, pstride[p]] block at code_off[p]
  const uint64_t* code_off;   // [nlist] byte offset of the partition block
  const uint32_t* plen;       // [nlist] rows of p kept on this handle (0 = not owned / empty)
  const uint32_t* pstride;    // [nlist] plen rounded up to 16
  const uint32_t* lrow0;      // [nlist] local row position of the partition's first row
  const uint64_t* grow0;      // [nlist] global index position of the partition's first row
  const uint64_t* row_ids;    // [n_local] or nullptr (identity: rowid = global position)
  const void* raw;            // [n_local, dim] or nullptr
  uint32_t raw_dtype;
  uint32_t raw_by_global;     // 1: `raw` is the caller's whole column [n_rows, dim] in GLOBAL index order
                              //    (MI355_INDEX_RAW_HOST_MAPPED), addressed by global position
};

// local row position -> global index position: the partition whose local range holds `pos`
// (lrow0 is non-decreasing; empty / foreign partitions have zero length)
__device__ __forceinline__ uint64_t global_pos_of(const IndexView& ix, uint32_t pos) {
  uint32_t lo = 0, hi = ix.nlist;  // last p with lrow0[p] <= pos and plen[p] > 0 covering pos
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (ix.lrow0[mid] <= pos) lo = mid; else hi = mid;
  }
  // partitions sharing the same lrow0 (zero-length ones) sit before the owner: step to the one that has rows
  while (lo > 0 && ix.plen[lo] == 0) --lo;
  return ix.grow0[lo] + (pos - ix.lrow0[lo]);
}

// ------------------------------------------------------------------ K0 -----
// One wave per query.  The chains of the contract are sequential in d, so one lane
// runs them — but from LDS in 16-B reads, after the wave has copied the query in
// coalesced (one thread per query with scalar global loads was latency-bound: 149 us
// per 2048-query batch, a visible slice of a multi-GPU step).
static __global__ __launch_bounds__(256) void k_prep_queries(const float* __restrict__ q, uint32_t nq, uint32_t dim,
                                                      uint32_t metric, float* __restrict__ qp,
                                                      float* __restrict__ qq) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const uint32_t b = blockIdx.x * 4 + wid;
  const uint32_t dimp = (dim + 3u) & ~3u;
  float* sq = (float*)smem + (size_t)wid * dimp;
  if (b >= nq) return;  // whole waves leave together; no block barrier below
  const float* src = q + (size_t)b * dim;
  float* dst = qp + (size_t)b * dim;
  for (uint32_t d = lane; d < dimp; d += 64) sq[d] = d < dim ? src[d] : 0.f;
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  auto chain_sq = [&]() -> float {  // sum of squares, d ascending (fma(0,0,acc) of the padding is exact)
    float acc = 0.f;
    for (uint32_t d = 0; d < dimp; d += 4) {
      const float4 v = *(const float4*)(sq + d);
      acc = __fmaf_rn(v.x, v.x, acc);
      acc = __fmaf_rn(v.y, v.y, acc);
      acc = __fmaf_rn(v.z, v.z, acc);
      acc = __fmaf_rn(v.w, v.w, acc);
    }
    return acc;
  };
  float acc = 0.f;
  if (lane == 0) acc = chain_sq();
  acc = readlane_f(acc, 0);
  if (metric == MI355_METRIC_COSINE) {
    const float nrm = ieee_sqrtf(acc);
    __builtin_amdgcn_wave_barrier();
    for (uint32_t d = lane; d < dim; d += 64) {
      const float v = ieee_divf(sq[d], nrm);
      sq[d] = v;
      dst[d] = v;
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    float acc2 = 0.f;
    if (lane == 0) acc2 = chain_sq();
    if (lane == 0) qq[b] = acc2;
  } else {
    for (uint32_t d = lane; d < dim; d += 64) dst[d] = sq[d];
    if (lane == 0) qq[b] = acc;
  }
}

static __global__ void k_centroid_norms(const float* __restrict__ c, uint32_t nlist, uint32_t dim,
                                 float* __restrict__ cn) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nlist) return;
  const float* src = c + (size_t)p * dim;
  float acc = 0.f;
  for (uint32_t d = 0; d < dim; ++d) acc = __fmaf_rn(src[d], src[d], acc);
  cn[p] = acc;
}

// ------------------------------------------------------------------ K1 -----
// Register-tiled f32 "GEMM" whose every accumulator is the d-ascending fma
// chain of the contract.  64 queries x 64 centroids per 256-thread block,
// 4x4 per thread, K staged 16 at a time through LDS (zero padding is exact:
// fma(0,0,acc) == acc for the +0-started chain).
#define CO_T 64
#define CO_K 16
static __global__ __launch_bounds__(256) void k_coarse_tile(
    const float* __restrict__ qp, const float* __restrict__ qq, uint32_t nq,
    const float* __restrict__ cen, const float* __restrict__ cn, uint32_t nlist, uint32_t dim,
    uint32_t metric, float* __restrict__ out /*[nq, nlist]*/) {
  __shared__ float sq[CO_K][CO_T + 4];
  __shared__ float sc[CO_K][CO_T + 4];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const uint32_t q0 = blockIdx.y * CO_T, c0 = blockIdx.x * CO_T;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (uint32_t k0 = 0; k0 < dim; k0 += CO_K) {
    // 64 rows x 16 k per operand = 1024 elements, 4 per thread
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int idx = tid + e * 256;
      int r = idx >> 4, kk = idx & 15;
      uint32_t k = k0 + kk;
      float vq = 0.f, vc = 0.f;
      if (k < dim) {
        if (q0 + r < nq) vq = qp[(size_t)(q0 + r) * dim + k];
        if (c0 + r < nlist) vc = cen[(size_t)(c0 + r) * dim + k];
      }
      sq[kk][r] = vq;
      sc[kk][r] = vc;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < CO_K; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = sq[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = sc[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __fmaf_rn(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t qi = q0 + ty * 4 + i;
    if (qi >= nq) continue;
    float qn = qq[qi];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t ci = c0 + tx * 4 + j;
      if (ci >= nlist) continue;
      float v;
      if (metric == MI355_METRIC_DOT)
        v = 1.0f - acc[i][j];
      else
        v = __fmaf_rn(-2.0f, acc[i][j], qn + cn[ci]);
      out[(size_t)qi * nlist + ci] = v;
    }
  }
}

// ------------------------------------------------------------------ K1 (MFMA)
// The same q x centroid products on the matrix cores: v_mfma_f32_32x32x2_f32 takes
// f32 inputs and accumulates D = C + a0*b0 + a1*b1 as two chained fmas, i.e. it IS
// the contract's d-ascending fmaf chain (bit-identical to k_coarse_tile and to the
// oracle; tests compare the distances with ==), at the f32 matrix rate (157 TFLOP/s,
// 1/16 of bf16).  256 threads = 4 waves as 2 x 2, each wave one 32 x 32 tile of a
// 64 queries x 64 centroids block; K staged 32 at a time through LDS (rows padded to
// 33 floats: the per-lane reads of one column hit 32 different banks).
#define CM_T 64
#define CM_K 32
typedef __attribute__((ext_vector_type(16))) float cm_f32x16;
typedef __attribute__((ext_vector_type(4))) float cm_f32x4;
static __global__ __launch_bounds__(256) void k_coarse_mfma(
    const float* __restrict__ qp, const float* __restrict__ qq, uint32_t nq,
    const float* __restrict__ cen, const float* __restrict__ cn, uint32_t nlist, uint32_t dim,
    uint32_t metric, float* __restrict__ out /*[nq, nlist]*/, ActiveMask act = ActiveMask()) {
  __shared__ float sa[CM_T][CM_K + 1];
  __shared__ float sb[CM_T][CM_K + 1];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const uint32_t q0 = blockIdx.y * CM_T, c0 = blockIdx.x * CM_T;
  if (!act.on(q0)) return;  // a query tile past the device-side batch size
  cm_f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int fi = lane & 31, fk = lane >> 5;
  // staging: 64 rows x 32 k per operand = 512 float4, 2 per thread per operand; the next
  // stage's loads are issued before this stage's MFMAs (register prefetch)
  const bool vec = (dim & 3u) == 0;
  float4 ra[2], rb[2];
  auto fetch = [&](uint32_t k0) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = tid + e * 256;
      const int r = idx >> 3, k4 = (idx & 7) * 4;
      const uint32_t k = k0 + k4;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
      if (vec) {
        if (k < dim) {
          if (q0 + r < nq) va = *(const float4*)(qp + (size_t)(q0 + r) * dim + k);
          if (c0 + r < nlist) vb = *(const float4*)(cen + (size_t)(c0 + r) * dim + k);
        }
      } else {
        float* pa = (float*)&va;
        float* pb = (float*)&vb;
        for (int t = 0; t < 4; ++t)
          if (k + t < dim) {
            if (q0 + r < nq) pa[t] = qp[(size_t)(q0 + r) * dim + k + t];
            if (c0 + r < nlist) pb[t] = cen[(size_t)(c0 + r) * dim + k + t];
          }
      }
      ra[e] = va;
      rb[e] = vb;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = tid + e * 256;
      const int r = idx >> 3, k4 = (idx & 7) * 4;
      sa[r][k4 + 0] = ra[e].x;
      sa[r][k4 + 1] = ra[e].y;
      sa[r][k4 + 2] = ra[e].z;
      sa[r][k4 + 3] = ra[e].w;
      sb[r][k4 + 0] = rb[e].x;
      sb[r][k4 + 1] = rb[e].y;
      sb[r][k4 + 2] = rb[e].z;
      sb[r][k4 + 3] = rb[e].w;
    }
  };
  fetch(0);
  for (uint32_t k0 = 0; k0 < dim; k0 += CM_K) {
    stash();
    __syncthreads();
    if (k0 + CM_K < dim) fetch(k0 + CM_K);
#pragma unroll
    for (int kk = 0; kk < CM_K; kk += 2) {
      const float a = sa[wr * 32 + fi][kk + fk];
      const float b = sb[wc * 32 + fi][kk + fk];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // D: col = lane & 31 (centroid), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (query)
  const uint32_t ci = c0 + wc * 32 + fi;
  const float cnv = ci < nlist ? cn[ci] : 0.f;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const uint32_t qi = q0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * fk;
    if (qi >= nq || ci >= nlist) continue;
    float v;
    if (metric == MI355_METRIC_DOT)
      v = 1.0f - acc[reg];
    else
      v = __fmaf_rn(-2.0f, acc[reg], qq[qi] + cnv);
    out[(size_t)qi * nlist + ci] = v;
  }
}

// The same kernel register-blocked 2 x 2: a wave owns a 64 x 64 block (four 32 x 32 MFMA tiles), the workgroup 128 queries x
// 128 centroids.  Every output element still accumulates its products k-ascending in ONE accumulator, so the bits are
// k_coarse_mfma's; per MFMA the LDS reads, the staging stores and the barriers are halved (the 64 x 64 kernel ran at 0.57-0.66
// of the f32 matrix peak).  Used when the launch still has two workgroups per CU (batches of ~1024 queries and up).
#define CM2_T 128
static __global__ __launch_bounds__(256) void k_coarse_mfma2(
    const float* __restrict__ qp, const float* __restrict__ qq, uint32_t nq,
    const float* __restrict__ cen, const float* __restrict__ cn, uint32_t nlist, uint32_t dim,
    uint32_t metric, float* __restrict__ out /*[nq, nlist]*/, ActiveMask act = ActiveMask()) {
  __shared__ float sa[CM2_T][CM_K + 1];
  __shared__ float sb[CM2_T][CM_K + 1];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const uint32_t q0 = blockIdx.y * CM2_T, c0 = blockIdx.x * CM2_T;
  if (!act.on(q0)) return;  // a query tile past the device-side batch size
  cm_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int fi = lane & 31, fk = lane >> 5;
  // staging: 128 rows x 32 k per operand = 1024 16-B pieces, 4 per thread per operand (dim % 4 == 0: the launch site checks)
  cm_f32x4 ra[4], rb[4];
  auto fetch = [&](uint32_t k0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * 256;
      const int r = idx >> 3, k4 = (idx & 7) * 4;
      const uint32_t k = k0 + k4;
      cm_f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = va;
      if (k < dim) {
        if (q0 + r < nq) va = *(const cm_f32x4*)(qp + (size_t)(q0 + r) * dim + k);
        if (c0 + r < nlist) vb = *(const cm_f32x4*)(cen + (size_t)(c0 + r) * dim + k);
      }
      ra[e] = va;
      rb[e] = vb;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * 256;
      const int r = idx >> 3, k4 = (idx & 7) * 4;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        sa[r][k4 + t] = ra[e][t];
        sb[r][k4 + t] = rb[e][t];
      }
    }
  };
  fetch(0);
  for (uint32_t k0 = 0; k0 < dim; k0 += CM_K) {
    stash();
    __syncthreads();
    if (k0 + CM_K < dim) fetch(k0 + CM_K);
#pragma unroll
    for (int kk = 0; kk < CM_K; kk += 2) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = sa[wr * 64 + i * 32 + fi][kk + fk];
        b[i] = sb[wc * 64 + i * 32 + fi][kk + fk];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // D: col = lane & 31 (centroid), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (query)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const uint32_t ci = c0 + wc * 64 + j * 32 + fi;
    const float cnv = ci < nlist ? cn[ci] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const uint32_t qi = q0 + wr * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * fk;
        if (qi >= nq || ci >= nlist) continue;
        float v;
        if (metric == MI355_METRIC_DOT)
          v = 1.0f - acc[i][j][reg];
        else
          v = __fmaf_rn(-2.0f, acc[i][j][reg], qq[qi] + cnv);
        out[(size_t)qi * nlist + ci] = v;
      }
  }
}

// ------------------------------------------------------------------ K0+K1 (latency mode) ----
// Up to CS_MAXQ queries: prep and coarse in ONE launch, the centroid matrix streamed by nlist / 64
// single-wave workgroups (lane = centroid, its row read in 16-B pieces, sixteen in flight), every
// accumulator the contract's d-ascending fmaf chain (bit-identical to k_coarse_mfma / k_coarse_tile).
// Each workgroup prepares the queries for itself in LDS (cosine normalisation, |q|^2 chains: 1.3 us
// of serial fmas per query, one lane each); workgroup 0 also writes qp / qq for the scan.  The
// MFMA kernel at one query is 64 workgroups x 24 barrier-bound k-steps: 45 us; this is ~8 us.
#define CS_MAXQ 8
static __global__ __launch_bounds__(64) void k_coarse_small(const float* __restrict__ q, uint32_t nq, uint32_t dim, uint32_t metric,
                                                            const float* __restrict__ cen, const float* __restrict__ cn, uint32_t nlist,
                                                            float* __restrict__ qp, float* __restrict__ qq_out,
                                                            float* __restrict__ out /*[nq, nlist]*/) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t dimp = (dim + 3u) & ~3u;
  float* sq = (float*)smem;                 // [nq][dimp]
  float* sqq = sq + (size_t)nq * dimp;      // [nq]
  const int lane = threadIdx.x;
  for (uint32_t j = 0; j < nq; ++j)
    for (uint32_t d = lane; d < dimp; d += 64) sq[(size_t)j * dimp + d] = d < dim ? q[(size_t)j * dim + d] : 0.f;
  __syncthreads();
  auto chain_sq = [&](const float* v) -> float {  // sum of squares, d ascending (fma(0,0,acc) of the padding is exact)
    float acc = 0.f;
    for (uint32_t d = 0; d < dimp; d += 4) {
      const float4 x = *(const float4*)(v + d);
      acc = __fmaf_rn(x.x, x.x, acc);
      acc = __fmaf_rn(x.y, x.y, acc);
      acc = __fmaf_rn(x.z, x.z, acc);
      acc = __fmaf_rn(x.w, x.w, acc);
    }
    return acc;
  };
  if ((uint32_t)lane < nq) sqq[lane] = chain_sq(sq + (size_t)lane * dimp);
  __syncthreads();
  if (metric == MI355_METRIC_COSINE) {
    for (uint32_t j = 0; j < nq; ++j) {
      const float nrm = ieee_sqrtf(sqq[j]);
      for (uint32_t d = lane; d < dim; d += 64) sq[(size_t)j * dimp + d] = ieee_divf(sq[(size_t)j * dimp + d], nrm);
    }
    __syncthreads();
    if ((uint32_t)lane < nq) sqq[lane] = chain_sq(sq + (size_t)lane * dimp);
    __syncthreads();
  }
  if (blockIdx.x == 0) {
    for (uint32_t j = 0; j < nq; ++j)
      for (uint32_t d = lane; d < dim; d += 64) qp[(size_t)j * dim + d] = sq[(size_t)j * dimp + d];
    if ((uint32_t)lane < nq) qq_out[lane] = sqq[lane];
  }
  const uint32_t c = blockIdx.x * 64u + lane;
  if (c >= nlist) return;
  const float* row = cen + (size_t)c * dim;
  float acc[CS_MAXQ];
#pragma unroll
  for (int j = 0; j < CS_MAXQ; ++j) acc[j] = 0.f;
  if ((dim & 3u) == 0) {
    constexpr int PF = 16;  // 16-B pieces in flight per lane: the kernel is one wave per 64 centroids, latency-bound
    for (uint32_t d0 = 0; d0 < dim; d0 += 4 * PF) {
      float4 r[PF];
#pragma unroll
      for (int u = 0; u < PF; ++u) r[u] = *(const float4*)(row + min(d0 + 4u * u, dim - 4u));  // (clamped, unconditional: see k_coarse_lat)
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        if (d0 + 4 * u >= dim) break;
#pragma unroll
        for (int j = 0; j < CS_MAXQ; ++j) {
          if ((uint32_t)j < nq) {
            const float4 x = *(const float4*)(sq + (size_t)j * dimp + d0 + 4 * u);  // broadcast
            acc[j] = __fmaf_rn(x.x, r[u].x, acc[j]);
            acc[j] = __fmaf_rn(x.y, r[u].y, acc[j]);
            acc[j] = __fmaf_rn(x.z, r[u].z, acc[j]);
            acc[j] = __fmaf_rn(x.w, r[u].w, acc[j]);
          }
        }
      }
    }
  } else {
    for (uint32_t d = 0; d < dim; ++d) {
      const float r = row[d];
#pragma unroll
      for (int j = 0; j < CS_MAXQ; ++j)
        if ((uint32_t)j < nq) acc[j] = __fmaf_rn(sq[(size_t)j * dimp + d], r, acc[j]);
    }
  }
  const float cnv = cn[c];
#pragma unroll
  for (int j = 0; j < CS_MAXQ; ++j) {
    if ((uint32_t)j < nq) {
      float v;
      if (metric == MI355_METRIC_DOT)
        v = 1.0f - acc[j];
      else
        v = __fmaf_rn(-2.0f, acc[j], sqq[j] + cnv);
      out[(size_t)j * nlist + c] = v;
    }
  }
}

// The same launch for centroid tables too small to give every CU a wave at 64 centroids per wave (nlist 4096 is
// 64 waves on 256 CUs, each waiting for twelve dependent rounds of loads: 40 us): LPC lanes share a centroid.  They
// load its row in 16-B pieces side by side (16 * LPC contiguous bytes per load instruction and centroid), stage a
// round of PF * LPC pieces in LDS and each lane then walks the whole round in d order for ITS queries (query j is
// lane j % LPC's), so every accumulator is still the one d-ascending chain.  LPC times the waves = LPC times the
// bytes in flight; the chains themselves take what they took.
// (register arrays of the small-batch coarse kernels are clang ext-vectors, not HIP's float4 struct: an array of 12+
//  float4 stays in SCRATCH — hipcc then waits for every global load on its own to store it there: 24 serial HBM round
//  trips, 16 of k_coarse_lat's 27 us; scripts/check_scratch.py lists the kernels that use scratch)
typedef __attribute__((ext_vector_type(4))) float cl_f32x4;
template <int LPC>
static __global__ __launch_bounds__(64) void k_coarse_split(const float* __restrict__ q, uint32_t nq, uint32_t dim, uint32_t metric,
                                                            const float* __restrict__ cen, const float* __restrict__ cn, uint32_t nlist,
                                                            float* __restrict__ qp, float* __restrict__ qq_out,
                                                            float* __restrict__ out /*[nq, nlist]*/) {
  constexpr int CPW = 64 / LPC;                      // centroids per wave
  constexpr int PF = 16;                             // pieces in flight per lane
  constexpr int RP = PF * LPC;                       // pieces of a row per round
  constexpr int JPL = (CS_MAXQ + LPC - 1) / LPC;     // queries per lane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t dimq = dim + 4u;                    // dim % 4 == 0 here; the pad staggers the queries' banks
  float* sq = (float*)smem;                          // [nq][dimq]
  float* sqq = sq + (size_t)nq * dimq;               // [CS_MAXQ]
  cl_f32x4* stage = (cl_f32x4*)(sqq + CS_MAXQ);          // [CPW][RP + 1]
  const int lane = threadIdx.x;
  for (uint32_t j = 0; j < nq; ++j)
    for (uint32_t d = lane; d < dim; d += 64) sq[(size_t)j * dimq + d] = q[(size_t)j * dim + d];
  __syncthreads();
  auto chain_sq = [&](const float* v) -> float {
    float acc = 0.f;
    for (uint32_t d = 0; d < dim; d += 4) {
      const cl_f32x4 x = *(const cl_f32x4*)(v + d);
      acc = __fmaf_rn(x.x, x.x, acc);
      acc = __fmaf_rn(x.y, x.y, acc);
      acc = __fmaf_rn(x.z, x.z, acc);
      acc = __fmaf_rn(x.w, x.w, acc);
    }
    return acc;
  };
  if ((uint32_t)lane < nq) sqq[lane] = chain_sq(sq + (size_t)lane * dimq);
  __syncthreads();
  if (metric == MI355_METRIC_COSINE) {
    for (uint32_t j = 0; j < nq; ++j) {
      const float nrm = ieee_sqrtf(sqq[j]);
      for (uint32_t d = lane; d < dim; d += 64) sq[(size_t)j * dimq + d] = ieee_divf(sq[(size_t)j * dimq + d], nrm);
    }
    __syncthreads();
    if ((uint32_t)lane < nq) sqq[lane] = chain_sq(sq + (size_t)lane * dimq);
    __syncthreads();
  }
  if (blockIdx.x == 0) {
    for (uint32_t j = 0; j < nq; ++j)
      for (uint32_t d = lane; d < dim; d += 64) qp[(size_t)j * dim + d] = sq[(size_t)j * dimq + d];
    if ((uint32_t)lane < nq) qq_out[lane] = sqq[lane];
  }
  const uint32_t cl = (uint32_t)lane / LPC, sub = (uint32_t)lane % LPC;
  const uint32_t c = blockIdx.x * CPW + cl;
  const cl_f32x4* row = (const cl_f32x4*)(cen + (size_t)min(c, nlist - 1u) * dim);
  cl_f32x4* mine = stage + (size_t)cl * (RP + 1);
  const uint32_t np = dim / 4u;
  float acc[JPL];
  const float* qv[JPL];
  bool on[JPL];
#pragma unroll
  for (int i = 0; i < JPL; ++i) {
    const uint32_t j = sub + (uint32_t)i * LPC;
    acc[i] = 0.f;
    on[i] = j < nq;
    qv[i] = sq + (size_t)(on[i] ? j : 0u) * dimq;
  }
  for (uint32_t p0 = 0; p0 < np; p0 += RP) {
    // (unconditional loads at a clamped piece: guarded per array element, hipcc waits for every load on its own —
    //  see k_coarse_lat; the pieces past the row's end land in stage slots the chain never reads)
    cl_f32x4 r[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) r[u] = row[min(p0 + (uint32_t)u * LPC + sub, np - 1u)];
    __syncthreads();  // the previous round has been read
#pragma unroll
    for (int u = 0; u < PF; ++u) mine[u * LPC + sub] = r[u];
    __syncthreads();
    const uint32_t lim = min((uint32_t)RP, np - p0);
    if (on[0]) {
      for (uint32_t pp = 0; pp < lim; pp += 4) {  // np % 4 need not be 0: the inner bound is checked
        cl_f32x4 v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (pp + e < lim) v[e] = mine[pp + e];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (pp + e >= lim) break;
#pragma unroll
          for (int i = 0; i < JPL; ++i) {
            if (on[i]) {
              const cl_f32x4 x = *(const cl_f32x4*)(qv[i] + 4u * (p0 + pp + e));
              acc[i] = __fmaf_rn(x.x, v[e].x, acc[i]);
              acc[i] = __fmaf_rn(x.y, v[e].y, acc[i]);
              acc[i] = __fmaf_rn(x.z, v[e].z, acc[i]);
              acc[i] = __fmaf_rn(x.w, v[e].w, acc[i]);
            }
          }
        }
      }
    }
  }
  if (c >= nlist) return;
  const float cnv = cn[c];
#pragma unroll
  for (int i = 0; i < JPL; ++i) {
    if (on[i]) {
      const uint32_t j = sub + (uint32_t)i * LPC;
      float v;
      if (metric == MI355_METRIC_DOT)
        v = 1.0f - acc[i];
      else
        v = __fmaf_rn(-2.0f, acc[i], sqq[j] + cnv);
      out[(size_t)j * nlist + c] = v;
    }
  }
}
// ---- latency front, first half (dim % 4 == 0): the dot chains alone ---------------------------------------
// k_coarse_split makes every workgroup run the queries' |q|^2 chains before it touches a centroid — 768 dependent fmas
// fed by dependent LDS reads, ~13 us of the 39 us a single query's coarse stage took on the C3 index — although only
// the LAST step of a score, fma(-2, dot, |q|^2 + |c|^2), needs them.  Here the centroid workgroups write the raw dot
// chains (the same d-ascending chains, LPC lanes sharing a centroid's loads, one round of PF pieces per lane in flight),
// ONE extra workgroup runs the queries' own chains beside them — eight 16-B LDS reads ahead of 32 fmas, per query lane —
// writes qp / qq for the scan and arms the call's deadline word (what k_arm_deadline did in a launch of its own), and
// k_select_plan applies the last step when it builds its keys.
// (register arrays of this kernel are clang ext-vectors, not HIP's cl_f32x4 struct: an array of 12+ cl_f32x4 stays in SCRATCH —
//  hipcc then waits for every global load on its own to store it there: 24 serial HBM round trips, 16 of the kernel's 27 us)
typedef __attribute__((ext_vector_type(4))) float cl_f32x4;
// A single host query travels in the kernel's ARGUMENT block (3.5 KB of the 4 KB HIP allows): the runtime copies the
// arguments into the kernarg segment at launch anyway, so the query needs no staging copy, no copy kernel (4 us) and no
// launch boundary of its own; the workgroups read it with ordinary loads from the kernarg segment.
#define CL_ARG_FLOATS 896u
struct CoarseLatQuery {
  float v[CL_ARG_FLOATS];
};
template <int LPC, int PF>
static __global__ __launch_bounds__(64) void k_coarse_lat(CoarseLatQuery qarg, uint32_t q_in_arg, const float* __restrict__ q, uint32_t nq, uint32_t dim,
                                                          const float* __restrict__ cen, uint32_t nlist,
                                                          float* __restrict__ qp, float* __restrict__ qq_out,
                                                          float* __restrict__ out /*[nq, nlist] raw dot chains*/,
                                                          DevCtl* ctl, unsigned long long arm_ticks, uint32_t arm_reset, uint32_t cosine) {
  constexpr int CPW = 64 / LPC;                      // centroids per wave
  constexpr int RP = PF * LPC;                       // pieces of a row per round
  constexpr int JPL = (CS_MAXQ + LPC - 1) / LPC;     // queries per lane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t dimq = dim + 4u;                    // the pad staggers the queries' banks
  float* sq = (float*)smem;                          // [nq][dimq]
  cl_f32x4* stage = (cl_f32x4*)(sq + (size_t)nq * dimq); // [CPW][RP + 1]
  float* s_n = (float*)(stage + (size_t)CPW * (RP + 1)); // [CS_MAXQ] raw |q|^2 (cosine)
  const int lane = threadIdx.x;
  const uint32_t np = dim / 4u;
#ifdef MI355_DEV_FRONT  // dev: block 0's stage times -> DevCtl::dev[0..3] (query to LDS / row loads + staging / chains / store)
  const unsigned long long cf_t0 = wall_clock64();
#endif
  if (q_in_arg) {  // (nq == 1, dim <= CL_ARG_FLOATS)
    const cl_f32x4* qa = (const cl_f32x4*)qarg.v;
    for (uint32_t d4 = lane; d4 < np; d4 += 64) *(cl_f32x4*)(sq + 4u * d4) = qa[d4];
  } else {
    for (uint32_t j = 0; j < nq; ++j)
      for (uint32_t d4 = lane; d4 < np; d4 += 64) *(cl_f32x4*)(sq + (size_t)j * dimq + 4u * d4) = *(const cl_f32x4*)(q + (size_t)j * dim + 4u * d4);
  }
  __syncthreads();
  // one query lane's |v|^2 chain over an LDS row: eight 16-B reads ahead of 32 fmas
  auto chain_sq = [&](const cl_f32x4* v) -> float {
    float acc = 0.f;
    uint32_t p0 = 0;
    for (; p0 + 8 <= np; p0 += 8) {
      cl_f32x4 x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = v[p0 + e];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        acc = __fmaf_rn(x[e].x, x[e].x, acc);
        acc = __fmaf_rn(x[e].y, x[e].y, acc);
        acc = __fmaf_rn(x[e].z, x[e].z, acc);
        acc = __fmaf_rn(x[e].w, x[e].w, acc);
      }
    }
    for (; p0 < np; ++p0) {
      const cl_f32x4 x = v[p0];
      acc = __fmaf_rn(x.x, x.x, acc);
      acc = __fmaf_rn(x.y, x.y, acc);
      acc = __fmaf_rn(x.z, x.z, acc);
      acc = __fmaf_rn(x.w, x.w, acc);
    }
    return acc;
  };
  if (cosine) {  // every workgroup normalises its own copy: q / sqrt(|q|^2), exactly k_coarse_small's arithmetic (+4 us)
    if ((uint32_t)lane < nq) s_n[lane] = chain_sq((const cl_f32x4*)(sq + (size_t)lane * dimq));
    __syncthreads();
    for (uint32_t j = 0; j < nq; ++j) {
      const float nrm = ieee_sqrtf(s_n[j]);
      for (uint32_t d = lane; d < dim; d += 64) sq[(size_t)j * dimq + d] = ieee_divf(sq[(size_t)j * dimq + d], nrm);
    }
    __syncthreads();
  }
  if (blockIdx.x == gridDim.x - 1u) {  // the queries' own workgroup
    if (lane == 0 && ctl) {
      ctl->deadline = arm_ticks ? (unsigned long long)wall_clock64() + arm_ticks : 0ull;
      ctl->timed_out = 0;
      if (arm_reset) {
        ctl->rows_scanned = 0ull;
        ctl->short_queries = 0u;
      }
    }
    for (uint32_t j = 0; j < nq; ++j)
      for (uint32_t d4 = lane; d4 < np; d4 += 64) *(cl_f32x4*)(qp + (size_t)j * dim + 4u * d4) = *(const cl_f32x4*)(sq + (size_t)j * dimq + 4u * d4);
    if ((uint32_t)lane < nq) qq_out[lane] = chain_sq((const cl_f32x4*)(sq + (size_t)lane * dimq));
    return;
  }
#ifdef MI355_DEV_FRONT
  const unsigned long long cf_t1 = wall_clock64();
  unsigned long long cf_t2 = 0;
#endif
  const uint32_t cl = (uint32_t)lane / LPC, sub = (uint32_t)lane % LPC;
  const uint32_t c = blockIdx.x * CPW + cl;
  const cl_f32x4* row = (const cl_f32x4*)(cen + (size_t)min(c, nlist - 1u) * dim);
  cl_f32x4* mine = stage + (size_t)cl * (RP + 1);
  float acc[JPL];
  const float* qv[JPL];
  bool on[JPL];
#pragma unroll
  for (int i = 0; i < JPL; ++i) {
    const uint32_t j = sub + (uint32_t)i * LPC;
    acc[i] = 0.f;
    on[i] = j < nq;
    qv[i] = sq + (size_t)(on[i] ? j : 0u) * dimq;
  }
  for (uint32_t p0 = 0; p0 < np; p0 += RP) {
    // one round: PF pieces per lane, all in flight at once — unconditional loads at a clamped index (see above); the
    // pieces past the row's end land in stage slots the chain never reads
    cl_f32x4 r[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) r[u] = row[min(p0 + (uint32_t)u * LPC + sub, np - 1u)];
    if (p0) __syncthreads();  // the previous round has been read
#pragma unroll
    for (int u = 0; u < PF; ++u) mine[u * LPC + sub] = r[u];
    __syncthreads();
#ifdef MI355_DEV_FRONT
    if (p0 == 0) cf_t2 = wall_clock64();
#endif
    const uint32_t lim = min((uint32_t)RP, np - p0);
    if (on[0]) {
      // Eight pieces per step: the row's and the queries' LDS reads of a step are all issued before its 32 fmas per
      // query — read inside the chain, every fma group waited for an LDS round trip of its own.
      uint32_t pp = 0;
      for (; pp + 8 <= lim; pp += 8) {
        cl_f32x4 v[8], x[JPL][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = mine[pp + e];
#pragma unroll
          for (int i = 0; i < JPL; ++i) x[i][e] = *(const cl_f32x4*)(qv[i] + 4u * (p0 + pp + e));  // (an idle slot reads query 0)
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
          for (int i = 0; i < JPL; ++i) {
            acc[i] = __fmaf_rn(x[i][e].x, v[e].x, acc[i]);
            acc[i] = __fmaf_rn(x[i][e].y, v[e].y, acc[i]);
            acc[i] = __fmaf_rn(x[i][e].z, v[e].z, acc[i]);
            acc[i] = __fmaf_rn(x[i][e].w, v[e].w, acc[i]);
          }
        }
      }
      for (; pp < lim; ++pp) {
        const cl_f32x4 v = mine[pp];
#pragma unroll
        for (int i = 0; i < JPL; ++i) {
          const cl_f32x4 x = *(const cl_f32x4*)(qv[i] + 4u * (p0 + pp));
          acc[i] = __fmaf_rn(x.x, v.x, acc[i]);
          acc[i] = __fmaf_rn(x.y, v.y, acc[i]);
          acc[i] = __fmaf_rn(x.z, v.z, acc[i]);
          acc[i] = __fmaf_rn(x.w, v.w, acc[i]);
        }
      }
    }
  }
#ifdef MI355_DEV_FRONT
  const unsigned long long cf_t3 = wall_clock64();
#endif
  if (c >= nlist) return;
#pragma unroll
  for (int i = 0; i < JPL; ++i)
    if (on[i]) out[(size_t)(sub + (uint32_t)i * LPC) * nlist + c] = acc[i];
#ifdef MI355_DEV_FRONT
  if (ctl && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && lane == 0) {
    atomicAdd(&ctl->dev[0], (uint32_t)(cf_t1 - cf_t0));
    atomicAdd(&ctl->dev[1], (uint32_t)(cf_t2 - cf_t1));
    atomicAdd(&ctl->dev[2], (uint32_t)(cf_t3 - cf_t2));
    atomicAdd(&ctl->dev[3], 1u);
  }
#endif
}
template <int LPC, int PF>
static inline size_t coarse_lat_lds(uint32_t nq, uint32_t dim) {
  return (size_t)nq * (dim + 4u) * sizeof(float) + (size_t)(64 / LPC) * (PF * LPC + 1) * 16 + CS_MAXQ * sizeof(float);
}
static inline size_t coarse_split_lds(uint32_t nq, uint32_t dim, int lpc) {
  return ((size_t)nq * (dim + 4u) + CS_MAXQ) * sizeof(float) + (size_t)(64 / lpc) * (16 * lpc + 1) * 16;
}

// ------------------------------------------------------------------ K1b ----
// One 256-thread block per query: 4-pass byte radix select of the nprobe-th
// smallest coarse key, then emit {key < T} (any order) followed by the
// lowest-id {key == T} rows.  Works for any nlist (C4: 65536).  Also adds the
// query's probed rows to the stats counters.
static __global__ __launch_bounds__(256) void k_select_probes(
    const float* __restrict__ coarse, uint32_t nlist, uint32_t nprobe,
    const uint32_t* __restrict__ plen, uint32_t* __restrict__ probes /*[nq, nprobe]*/,
    unsigned long long* __restrict__ stat_rows, ActiveMask act = ActiveMask(), uint32_t* __restrict__ qthr = nullptr) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t s_prefix, s_need, s_less, s_wave_cnt[4], s_running, s_best_at, s_eq_all;
  __shared__ unsigned long long s_rows, s_best;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t b = blockIdx.x;
  const float* src = coarse + (size_t)b * nlist;
  uint32_t* out = probes + (size_t)b * nprobe;
  if (!act.on(b)) return;  // inactive slot: the planner, the scan and the merge skip it too
  if (tid == 0) {
    if (qthr) qthr[b] = 0xFFFFFFFFu;  // the query's running distance bound of the scan: none yet
    s_prefix = 0;
    s_need = nprobe;
    s_less = 0;
    s_running = 0;
    s_rows = 0;
    s_best = ~0ull;
    s_best_at = 0;
    s_eq_all = 0;
  }
  uint32_t mask = 0;
  for (int byte = 3; byte >= 0; --byte) {
    hist[tid] = 0;
    __syncthreads();
    uint32_t prefix = s_prefix;
    // (runs of one bin are counted in a register and added once: coarse distances share their top key byte, and at
    //  nlist 65536 the first pass was 65536 same-address LDS atomics per query — serialised)
    uint32_t cur = 0, run = 0;
    for (uint32_t p = tid; p < nlist; p += 256) {
      uint32_t key = f32_sort_key(src[p]);
      if ((key & mask) == prefix) {
        const uint32_t bin = (key >> (8 * byte)) & 255u;
        if (bin != cur && run) {
          atomicAdd(&hist[cur], run);
          run = 0;
        }
        cur = bin;
        ++run;
      }
    }
    if (run) atomicAdd(&hist[cur], run);
    __syncthreads();
    {
      // the bin holding the need-th smallest key: inclusive scan of the 256 counts over the 256 threads (a
      // serial walk by one thread was 4 x 256 dependent LDS reads — most of this kernel's single-query time)
      const uint32_t h = hist[tid];
      uint32_t inc = h;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, off);
        if (lane >= off) inc += v;
      }
      if (lane == 63) s_wave_cnt[wid] = inc;
      __syncthreads();
      uint32_t base = 0;
      for (int w = 0; w < wid; ++w) base += s_wave_cnt[w];
      inc += base;
      const uint32_t need = s_need;
      __syncthreads();  // every thread has read s_need / s_wave_cnt
      if (inc >= need && inc - h < need) {  // exactly one thread
        s_need = need - (inc - h);
        s_prefix = prefix | ((uint32_t)tid << (8 * byte));
        if (byte == 0) s_eq_all = (h == need - (inc - h)) ? 1u : 0u;  // every key equal to the threshold is taken
      }
    }
    mask |= 255u << (8 * byte);
    __syncthreads();
  }
  const uint32_t T = s_prefix;
  const uint32_t need_eq = s_need;           // rows with key == T to take (>= 1)
  const uint32_t n_less = nprobe - need_eq;  // rows with key < T
  unsigned long long rows = 0;
  // no tie is cut at the threshold (the usual case): the list is every key <= T in any order, no ranks to agree on
  const bool unordered = s_eq_all != 0u;
  for (uint32_t p = tid; unordered && p < nlist; p += 256) {
    const uint32_t key = f32_sort_key(src[p]);
    if (key <= T) {
      out[atomicAdd(&s_less, 1u)] = p;
      rows += plen[p];
      atomicMin(&s_best, ((unsigned long long)key << 32) | p);
    }
  }
  for (uint32_t p0 = 0; !unordered && p0 < nlist; p0 += 256) {
    uint32_t p = p0 + tid;
    uint32_t key = p < nlist ? f32_sort_key(src[p]) : 0xFFFFFFFFu;
    bool less = p < nlist && key < T;
    bool eq = p < nlist && key == T;
    if (less) {
      out[atomicAdd(&s_less, 1u)] = p;
      rows += plen[p];
      atomicMin(&s_best, ((unsigned long long)key << 32) | p);
    }
    // ordered rank among the equal keys (ascending partition id)
    uint64_t bal = __ballot(eq);
    if (lane == 0) s_wave_cnt[wid] = (uint32_t)__popcll((unsigned long long)bal);
    __syncthreads();
    uint32_t base = s_running;
    for (int w = 0; w < wid; ++w) base += s_wave_cnt[w];
    uint32_t rank = base + (uint32_t)__popcll((unsigned long long)(bal & ((1ull << lane) - 1ull)));
    if (eq && rank < need_eq) {
      out[n_less + rank] = p;
      rows += plen[p];
      atomicMin(&s_best, ((unsigned long long)key << 32) | p);
    }
    __syncthreads();
    if (tid == 0) s_running = base + s_wave_cnt[0] + s_wave_cnt[1] + s_wave_cnt[2] + s_wave_cnt[3];
    __syncthreads();
  }
  if (rows) atomicAdd(&s_rows, rows);
  __syncthreads();
  if (tid == 0 && stat_rows) atomicAdd(stat_rows, s_rows);
  // The NEAREST partition (ties: lowest id) moves to rank 0 — the order of the other ranks stays
  // arbitrary.  The scan's planner can then run every query's nearest partition first
  // (PlanArgs::best_first), so the query's distance bound exists before its other partitions are
  // scanned; the result set does not depend on the order.
  if (nprobe > 1) {
    const uint32_t best = (uint32_t)s_best;
    for (uint32_t i = tid; i < nprobe; i += 256)
      if (out[i] == best) s_best_at = i;
    __syncthreads();
    if (tid == 0 && s_best_at != 0) {
      out[s_best_at] = out[0];
      out[0] = best;
    }
  }
}

// two-phase search helpers (mi355_coarse_topn / mi355_search_probes)
// (partition id, distance) pairs of a slice's selected probes, in merge_topk's layout
static __global__ void k_emit_coarse_pairs(const uint32_t* __restrict__ probes, const float* __restrict__ coarse,
                                    uint32_t nq, uint32_t n_sel, uint32_t n_slice, uint32_t nprobe, uint32_t cent_lo,
                                    uint64_t* __restrict__ out_ids, float* __restrict__ out_dist,
                                    uint32_t* __restrict__ out_cnt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq * nprobe) return;
  const uint32_t b = i / nprobe, r = i % nprobe;
  if (r < n_sel) {
    const uint32_t p = probes[(size_t)b * n_sel + r];
    out_ids[i] = (uint64_t)cent_lo + p;
    out_dist[i] = coarse[(size_t)b * n_slice + p];
  } else {
    out_ids[i] = ~0ull;
    out_dist[i] = __builtin_huge_valf();
  }
  if (r == 0) out_cnt[b] = n_sel;
}

// external probe list (u64 ids) -> the u32 list the scan reads, plus the row counter
static __global__ void k_take_probes(const uint64_t* __restrict__ in, uint32_t n, uint32_t nlist,
                              const uint32_t* __restrict__ plen, uint32_t* __restrict__ out,
                              unsigned long long* __restrict__ stat_rows, uint32_t* __restrict__ bad,
                              uint32_t nprobe = 1, ActiveMask act = ActiveMask(), uint32_t* __restrict__ qthr = nullptr) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (!act.on(i / nprobe)) return;
  if (qthr && i % nprobe == 0u) qthr[i / nprobe] = 0xFFFFFFFFu;
  const uint64_t p = in[i];
  if (p >= nlist) {  // not a partition of this index: counted (mi355_stats.bad_probes; host-I/O calls fail) and
    atomicAdd(bad, 1u);  // turned into an EMPTY work item (the planner and the scan treat ids >= nlist as length 0)
    out[i] = 0xFFFFFFFFu;
    return;
  }
  out[i] = (uint32_t)p;
  if (stat_rows && plen[p]) atomicAdd(stat_rows, (unsigned long long)plen[p]);
}

// ------------------------------------------------------------ K2+K3+K4 -----
// SCAN_PAIR: one workgroup per (query, probe rank, slice of the partition).
//   1. residual r = q - c_p and the m x 256 f32 distance table, built straight
//      into LDS (never touches HBM);  LDS = m KiB + dim*4 B  (96 KiB + 3 KiB at C3)
//   2. ADC: each thread streams VPT consecutive rows per sub-quantiser row
//      (codes are sub-quantiser-major, so a wave reads VPT*64 contiguous bytes
//      per row: 1 KiB at VPT=16), gathers LUT[j][code] with ds_read_b32 and
//      accumulates j ascending with plain f32 adds (the contract's order)
//   3. per-wave exact top-kk (WaveTopK), merged across the block's waves
//      through the (now dead) LUT area, kk candidates per work item to HBM.
template <int VPT>
struct CodeVec;
template <>
struct CodeVec<4> {
  typedef uint32_t type;
};
template <>
struct CodeVec<16> {
  typedef uint4 type;
};

template <int VPT>
__device__ __forceinline__ uint32_t code_byte(const typename CodeVec<VPT>::type& v, int e);
template <>
__device__ __forceinline__ uint32_t code_byte<4>(const uint32_t& v, int e) {
  return (v >> (8 * e)) & 255u;
}
template <>
__device__ __forceinline__ uint32_t code_byte<16>(const uint4& v, int e) {
  uint32_t w = e < 4 ? v.x : e < 8 ? v.y : e < 12 ? v.z : v.w;
  return (w >> (8 * (e & 3))) & 255u;
}

struct ScanArgs {
  IndexView ix;
  const float* qp;          // [nq, dim] preprocessed queries
  const uint32_t* probes;   // [nq, nprobe]
  uint32_t nprobe;
  uint32_t slice_rows;      // rows per work item (multiple of 16)
  uint32_t n_slices;        // grid.x
  uint32_t kk;
  RangeFilter range;
  RowFilter filter;
  Cand* cand;               // [nq, nprobe, n_slices, kk]
  uint32_t dbg;             // dev ablation mask (MI355_DBG_SKIP): 1 LUT build, 2 ADC loop, 4 top-k
  DevCtl* ctl;              // deadline / counters of the call
  // SPILL (8-bit tables that exceed the LDS, e.g. dim 3072 with the default dim / 16 = 192 sub-vectors):
  // sub-quantisers [0, m_lds) keep their tables in LDS, the rest live in a per-work-item slice of
  // `lut_spill` ([grid blocks][m - m_lds][256] f32, served by L2) and are gathered from there
  uint32_t m_lds;
  float* lut_spill;
  ActiveMask act;           // device-side batch size (second pass of maximum_nprobes)
};

// rows per selection pass of a scan work item when kk exceeds the per-wave list capacity
#define SCAN_PASS_ROWS 256u

// (distance, rowid) floor of a selection pass, shared through LDS
struct PassFloor {
  float d;
  uint32_t on;
  uint64_t id;
};

__device__ __forceinline__ float finalize_dist(float acc, uint32_t metric, uint32_t m) {
  if (metric == MI355_METRIC_COSINE) return acc * 0.5f;
  if (metric == MI355_METRIC_DOT) return acc - (float)(m - 1);
  return acc;
}

// LR: per-wave list capacity in units of 64 entries (2: kk <= 64, 5: kk <= 256).
// NBITS: 8 (256-entry tables) or 4 (16-entry tables, two codes per byte: sub-quantiser 2t in the
// low nibble of byte t, 2t+1 in the high nibble; table/create_index.rs:96-101).
// MULTI: kk > 256 — the work item's rows are selected in passes of SCAN_PASS_ROWS: pass p keeps
// the best rows strictly above the last row of pass p-1 in the (distance, rowid) order and
// writes ranks p*256 .. of the item's kk output slots; the distance table is built once.
// SPILL: see ScanArgs::m_lds (the sum stays j-ascending: LDS part first, then the spilled part).
template <int VPT, int LR, int NTHREADS, int NBITS, bool MULTI, bool SPILL = false>
__global__ __launch_bounds__(NTHREADS) void k_scan_pair(ScanArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NW = NTHREADS / MI355_WAVE;
  constexpr uint32_t CB = 1u << NBITS;  // table entries per sub-quantiser
  const IndexView& ix = a.ix;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t s = blockIdx.x, r = blockIdx.y, b = blockIdx.z;
  if (!a.act.on(b)) return;
  if (ctl_expired(a.ctl)) return;  // block-uniform enough: a late block only wastes its own time
  Cand* out = a.cand + (((size_t)b * a.nprobe + r) * a.n_slices + s) * a.kk;
  const uint32_t p = a.probes[(size_t)b * a.nprobe + r];
  const uint32_t len = p < ix.nlist ? ix.plen[p] : 0u;  // out-of-range probe ids (search_probes) are empty items
  const uint32_t v0 = s * a.slice_rows;
  if (v0 >= len) {  // nothing here: mark the slot empty
    for (uint32_t g = tid; g < a.kk; g += NTHREADS) {
      Cand c;
      c.d = __builtin_huge_valf();
      c.pos = CAND_EMPTY_POS;
      c.id = ~0ull;
      out[g] = c;
    }
    return;
  }
  const uint32_t v1 = min(len, v0 + a.slice_rows);
  const uint32_t m_lds = SPILL ? a.m_lds : ix.m;
  const size_t lut_bytes = (size_t)m_lds * CB * 4;
  float* lut = (float*)smem;                  // [m_lds][CB]
  float* spill = nullptr;                     // [m - m_lds][CB] of this work item
  if (SPILL)
    spill = a.lut_spill + ((((size_t)b * gridDim.y + r) * gridDim.x + s) * (ix.m - m_lds)) * CB;
  float* res = (float*)(smem + lut_bytes);    // [dim]
  const float* q = a.qp + (size_t)b * ix.dim;

  // ---- K2: residual + distance table -------------------------------------
  if (ix.metric == MI355_METRIC_DOT) {
    for (uint32_t d = tid; d < ix.dim; d += NTHREADS) res[d] = q[d];
  } else {
    const float* c = ix.centroids + (size_t)p * ix.dim;
    for (uint32_t d = tid; d < ix.dim; d += NTHREADS) res[d] = q[d] - c[d];
  }
  __syncthreads();
  if (!(a.dbg & 1u)) {
    const uint32_t dsub = ix.dsub;
    for (uint32_t e = tid; e < ix.m * CB; e += NTHREADS) {
      const uint32_t j = e / CB;
      const float* cb = ix.codebook + (size_t)e * dsub;  // [j][c][*], coalesced over c
      const float* rj = res + j * dsub;
      float acc = 0.f;
      if (ix.metric == MI355_METRIC_DOT) {
        for (uint32_t t = 0; t < dsub; ++t) acc = __fmaf_rn(rj[t], cb[t], acc);
        acc = 1.0f - acc;
      } else {
        for (uint32_t t = 0; t < dsub; ++t) {
          float df = rj[t] - cb[t];
          acc = __fmaf_rn(df, df, acc);
        }
      }
      if (!SPILL || j < m_lds)
        lut[e] = acc;
      else
        spill[e - m_lds * CB] = acc;
    }
  }
  __syncthreads();  // (orders the block's global writes to `spill` before its reads, too)

  // ---- K3: ADC scan + K4 shuffle-free selection ----------------------------
  ListEnt* lists = (ListEnt*)(smem + lut_bytes + (((size_t)ix.dim * 4 + 15) & ~(size_t)15));
  uint32_t* s_cnt = (uint32_t*)(lists + (size_t)NW * LR * MI355_WAVE);  // [NW]
  PassFloor* s_floor = (PassFloor*)(((size_t)(s_cnt + NW) + 15) & ~(size_t)15);  // [1] (MULTI)
  const uint8_t* codes = ix.codes + ix.code_off[p];
  const uint32_t stride = ix.pstride[p];
  const uint32_t lrow0 = ix.lrow0[p];
  const uint64_t grow0 = ix.grow0[p];
  const uint64_t* rid = ix.row_ids;
  auto idof = [&](uint32_t pos) -> uint64_t { return rid ? rid[pos] : grow0 + (pos - lrow0); };
  const bool ranged = a.range.has_lower || a.range.has_upper;
  typedef typename CodeVec<VPT>::type cvec;

  for (uint32_t pass_base = 0;; pass_base += SCAN_PASS_ROWS) {
    const uint32_t kk_pass = MULTI ? min(a.kk - pass_base, SCAN_PASS_ROWS) : a.kk;
    bool fl_on = false;
    float fl_d = 0.f;
    uint64_t fl_id = 0;
    if (MULTI && pass_base) {
      fl_on = true;
      fl_d = s_floor->d;
      fl_id = s_floor->id;
    }
    WaveList<LR> wl;
    wl.init(lists + (size_t)wid * LR * MI355_WAVE, kk_pass);

    // The trip count is block-uniform: the body uses wave collectives (ballots),
    // so lanes past the end of the slice stay in the loop, re-read the slice's
    // first rows (always mapped) and are masked out of the selection.
    for (uint32_t base = v0; base < v1 && !(a.dbg & 2u); base += NTHREADS * VPT) {
      const uint32_t i0r = base + tid * VPT;
      const bool act = i0r < v1;
      const uint32_t i0 = act ? i0r : v0;
      float acc[VPT];
#pragma unroll
      for (int e = 0; e < VPT; ++e) acc[e] = 0.f;
      const uint8_t* col = codes + i0;
#pragma unroll 8
      for (uint32_t jb = 0; jb < (SPILL ? m_lds : ix.mb); ++jb) {
        cvec cv = *(const cvec*)(col + (size_t)jb * stride);
        if (NBITS == 8) {
          const float* t = lut + jb * 256;
#pragma unroll
          for (int e = 0; e < VPT; ++e) acc[e] = acc[e] + t[code_byte<VPT>(cv, e)];
        } else {  // sub-quantisers 2 jb (low nibble) then 2 jb + 1 (high nibble): j ascending
          const float* t0 = lut + (2 * jb) * 16;
          const float* t1 = t0 + 16;
#pragma unroll
          for (int e = 0; e < VPT; ++e) {
            const uint32_t by = code_byte<VPT>(cv, e);
            acc[e] = acc[e] + t0[by & 15u];
            acc[e] = acc[e] + t1[by >> 4];
          }
        }
      }
      if (SPILL) {  // the table's tail: 4-byte gathers from L2
#pragma unroll 4
        for (uint32_t jb = m_lds; jb < ix.mb; ++jb) {
          cvec cv = *(const cvec*)(col + (size_t)jb * stride);
          const float* t = spill + (size_t)(jb - m_lds) * 256;
#pragma unroll
          for (int e = 0; e < VPT; ++e) acc[e] = acc[e] + t[code_byte<VPT>(cv, e)];
        }
      }
      if (a.dbg & 4u) {  // ablation: keep the distances live, skip the selection
        float keep = 0.f;
#pragma unroll
        for (int e = 0; e < VPT; ++e) keep += acc[e];
        if (keep == -1.2345f) out[0].d = keep;
        continue;
      }
      // eligible rows: inside the slice, not NULL, inside the distance range (and above the pass floor)
      uint32_t elig = 0;
      float lmin = __builtin_huge_valf();
#pragma unroll
      for (int e = 0; e < VPT; ++e) {
        acc[e] = finalize_dist(acc[e], ix.metric, ix.m);
        bool ok = act && (i0r + e < v1) && (ranged ? in_range(acc[e], a.range) : acc[e] == acc[e]);
        if (MULTI && fl_on && ok) {
          ok = acc[e] > fl_d;
          if (acc[e] == fl_d) ok = idof(lrow0 + i0r + e) > fl_id;  // ties on the floor distance: rare
        }
        elig |= ok ? (1u << e) : 0u;
        lmin = ok ? fminf(lmin, acc[e]) : lmin;
      }
      // threshold: kk-th smallest lane minimum (>= kk rows of this wave are below it)
      float thr = wl.t_run;
      // (with a prefilter the lane minima include rows the filter may drop: only the list's own
      //  running threshold, built from permitted rows, is a valid bound)
      if (kk_pass <= MI355_WAVE && a.filter.mode == MI355_FILTER_NONE) {
        uint32_t tk = wave_kth_smallest_key(elig ? f32_sort_key(lmin) : 0xFFFFFFFFu, kk_pass);
        if (tk < 0xFF800000u) thr = fminf(thr, f32_from_sort_key(tk));  // below +inf's key
      }
      bool any = false;
#pragma unroll
      for (int e = 0; e < VPT; ++e) any |= ((elig >> e) & 1u) && acc[e] <= thr;
      if (__any(any)) {
#pragma unroll
        for (int e = 0; e < VPT; ++e) {
          bool take = ((elig >> e) & 1u) && acc[e] <= thr;
          if (a.filter.mode != MI355_FILTER_NONE && take) take = row_permitted(idof(lrow0 + i0r + e), a.filter);
          wl.append(take, acc[e], lrow0 + i0r + e, thr, lane, idof);
        }
      }
    }

    // ---- block result: exact kk_pass best of all waves' lists, written sorted -----
    if (wl.cnt > kk_pass) wl.compact(lane, idof);
    if (lane == 0) s_cnt[wid] = wl.cnt;
    __syncthreads();
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) total += s_cnt[w];
    const uint32_t n_out = min(total, kk_pass);
    for (uint32_t g = tid; g < (uint32_t)NW * kk_pass; g += NTHREADS) {
      const uint32_t w = g / kk_pass, j = g % kk_pass;
      if (j >= s_cnt[w]) continue;
      const ListEnt mine = lists[(size_t)w * LR * MI355_WAVE + j];
      uint32_t rank = 0;
      for (int w2 = 0; w2 < NW; ++w2) {
        const ListEnt* l2 = lists + (size_t)w2 * LR * MI355_WAVE;
        const uint32_t c2 = s_cnt[w2];
        for (uint32_t j2 = 0; j2 < c2; ++j2) {
          const ListEnt c = l2[j2];
          bool lt = c.d < mine.d;
          if (c.d == mine.d && c.pos != mine.pos) lt = idof(c.pos) < idof(mine.pos);
          rank += lt ? 1u : 0u;
        }
      }
      if (rank < kk_pass) {
        Cand o;
        o.d = mine.d;
        o.pos = mine.pos;
        o.id = idof(mine.pos);
        out[pass_base + rank] = o;
        if (MULTI && rank == kk_pass - 1) {  // the next pass starts strictly above this row
          s_floor->d = o.d;
          s_floor->id = o.id;
          s_floor->on = 1;
        }
      }
    }
    const bool more = MULTI && total >= kk_pass && pass_base + kk_pass < a.kk;
    if (!more) {
      for (uint32_t g = pass_base + n_out + tid; g < a.kk; g += NTHREADS) {
        Cand o;
        o.d = __builtin_huge_valf();
        o.pos = CAND_EMPTY_POS;
        o.id = ~0ull;
        out[g] = o;
      }
      break;
    }
    __syncthreads();  // the floor is published; the lists are rebuilt by the next pass
  }
}

// ------------------------------------------------------------------ K4 -----
// Per-query reducer: one wave scans the query's candidate slots and writes the k_out best in
// (distance, rowid) order; any k_out (k_out > 64 * KPL re-reads the slots once per pass of
// 64 * KPL rows, device_common.h wave_select_sorted).  Source s of query b starts at
// cand + s * src_stride + b * q_stride (scan output: [nq][n_src][kk]; gathered shard lists:
// [world][nq][kk]).
struct MergeArgs {
  const Cand* cand;
  uint32_t n_src, kk_in;
  uint64_t src_stride, q_stride;  // in Cand records
  const uint32_t* src_cnt;  // valid entries per (source, query) at src_cnt[s * cnt_stride + b], or nullptr
                            // (= kk_in, empties marked by pos)
  uint64_t cnt_stride;      // u32 words between two sources' count arrays
  uint64_t cnt_q_stride;    // u32 words between two queries' counts of one source (1: [source][query] arrays;
                            // the scan's per-item counts are [query][source]: cnt_stride 1, cnt_q_stride n_src)
  uint32_t nq;
  uint32_t k_out;     // rows written per query (k, or k*refine_factor)
  uint64_t* out_ids;  // [nq, k_out] or nullptr
  float* out_dist;    // [nq, k_out] or nullptr
  uint32_t* out_pos;  // [nq, k_out] or nullptr
  uint32_t* out_owner;  // [nq, k_out] source index of each winner, or nullptr
  Cand* out_cand;     // [nq, k_out] packed records (distance, pos, id), or nullptr
  uint32_t* out_cnt;  // [nq]
  const DevCtl* ctl;  // deadline flag (nullptr = none)
  ActiveMask act;     // device-side batch size (inactive queries are not merged)
};

// the scan's own layout: cand [nq][n_src][kk_in], empty slots marked by pos
static inline MergeArgs merge_args_dense(const Cand* cand, uint32_t n_src, uint32_t kk_in, uint32_t nq, uint32_t k_out) {
  MergeArgs m;
  m.cand = cand;
  m.n_src = n_src;
  m.kk_in = kk_in;
  m.src_stride = kk_in;
  m.q_stride = (uint64_t)n_src * kk_in;
  m.src_cnt = nullptr;
  m.cnt_stride = nq;
  m.cnt_q_stride = 1;
  m.nq = nq;
  m.k_out = k_out;
  m.out_ids = nullptr;
  m.out_dist = nullptr;
  m.out_pos = nullptr;
  m.out_owner = nullptr;
  m.out_cand = nullptr;
  m.out_cnt = nullptr;
  m.ctl = nullptr;
  m.act = ActiveMask();
  return m;
}

#define MERGE_PRE_SRC 1024u
#define MERGE_SHORT_CAP 8192u  // rows of the first short list (distance + slot, 8 B each in LDS)
#define MERGE_FINAL_CAP 2048u  // rows ranked against each other at the end
#define MERGE_BLOCK_LDS ((MERGE_SHORT_CAP * 2u + MERGE_FINAL_CAP * 5u + 256u + 8u) * 4u)
// k-th smallest (1-based, k <= n) of the n keys get(0..n-1) by a byte-wise radix select over the whole block: four
// passes of {256-bin LDS histogram, bin pick by wave 0}.  hist[256], st[2] (prefix, need) are LDS; every thread of
// the NT-thread block calls it.
template <int NT, typename Get>
__device__ __forceinline__ uint32_t block_kth_smallest_key(Get get, uint32_t n, uint32_t k, uint32_t* hist, uint32_t* st) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  if (tid == 0) {
    st[0] = 0;
    st[1] = k;
  }
  uint32_t mask = 0;
  for (int byte = 3; byte >= 0; --byte) {
    if (tid < 256u) hist[tid] = 0;
    __syncthreads();
    const uint32_t prefix = st[0], need = st[1];
    for (uint32_t i0 = 0; i0 < n; i0 += NT) {
      const uint32_t i = i0 + tid;
      const uint32_t key = i < n ? get(i) : 0u;
      bool mine = i < n && (key & mask) == prefix;
      const uint32_t bin = (key >> (8 * byte)) & 255u;
      // distances of one query share their leading bits: a wave's keys fall into one or two bins in the first
      // passes, so the two most common bins are counted once per wave, the rest lane by lane
      uint64_t rem = __ballot(mine);
      for (int round = 0; round < 2 && rem; ++round) {
        const int leader = __ffsll((unsigned long long)rem) - 1;
        const uint32_t lb = __shfl(bin, leader);
        const uint64_t same = __ballot(mine && bin == lb);
        if ((int)lane == leader) atomicAdd(&hist[lb], (uint32_t)__popcll((unsigned long long)same));
        if (bin == lb) mine = false;
        rem &= ~same;
      }
      if (mine) atomicAdd(&hist[bin], 1u);
    }
    __syncthreads();
    if (tid < 64u) {  // lane l owns bins 4l .. 4l+3
      const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
      const uint32_t sum = h0 + h1 + h2 + h3;
      uint32_t inc = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off);
        if ((int)lane >= off) inc += o;
      }
      const uint32_t exc = inc - sum;
      if (exc < need && need <= inc) {  // exactly one lane
        uint32_t bin = 4 * lane, before = exc;
        if (need > before + h0) {
          before += h0;
          ++bin;
          if (need > before + h1) {
            before += h1;
            ++bin;
            if (need > before + h2) {
              before += h2;
              ++bin;
            }
          }
        }
        st[0] = prefix | (bin << (8 * byte));
        st[1] = need - before;
      }
    }
    mask |= 255u << (8 * byte);
    __syncthreads();
  }
  return st[0];
}

// NW > 1 (a batch too small to fill the chip with single waves — the sliced single query's 512 work items all run at
// once, without a query bound, and every one returns its own kk best: 5120 .. 128 000 filled slots for ONE wave, whose
// selector pays ~1 us per row it admits).  The block of NW waves selects by keys instead (f32 sort key of the distance):
//   1. bound: when the query has more than MERGE_SHORT_CAP filled slots, the k_out-th smallest key of a sample of
//      64 * NW of them — at least k_out slots lie at or below it (two samples, the tighter bound wins: the first rows of
//      every source, and every (filled / 1024)-th filled slot);
//   2. sweep: all slots, independent loads; the ones at or below the bound go to a short list in LDS (distance, slot);
//   3. the exact k_out-th smallest key of the short list (radix select in LDS) cuts it to the final list: k_out rows
//      plus the ties of the last one;
//   4. the final list is ranked by counting in (distance, rowid) order — ids are unique, they are fetched where two
//      distances are equal — and the rows ranked below k_out are written where they belong.
// A list that overflows (more than MERGE_SHORT_CAP rows under the bound, more than MERGE_FINAL_CAP with the ties) or
// k_out > 64 * NW: wave 0 walks everything as the one-wave kernel does.
template <int KPL, int NW = 1>
__global__ __launch_bounds__(64 * NW) void k_merge_cands(MergeArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t b = blockIdx.x;
  if (a.ctl && a.ctl->timed_out) return;
  if (!a.act.on(b)) return;
  const Cand* src = a.cand + (size_t)b * a.q_stride;
  const uint32_t n = a.n_src * a.kk_in;
  uint64_t* oi = a.out_ids ? a.out_ids + (size_t)b * a.k_out : nullptr;
  float* od = a.out_dist ? a.out_dist + (size_t)b * a.k_out : nullptr;
  uint32_t* op = a.out_pos ? a.out_pos + (size_t)b * a.k_out : nullptr;
  uint32_t* oo = a.out_owner ? a.out_owner + (size_t)b * a.k_out : nullptr;
  Cand* oc = a.out_cand ? a.out_cand + (size_t)b * a.k_out : nullptr;
  for (uint32_t g = tid; g < a.k_out; g += MI355_WAVE * NW) {
    if (oi) oi[g] = ~0ull;
    if (od) od[g] = __builtin_huge_valf();
    if (op) op[g] = CAND_EMPTY_POS;
    if (oo) oo[g] = 0xFFFFFFFFu;
    if (oc) {
      Cand e;
      e.d = __builtin_huge_valf();
      e.pos = CAND_EMPTY_POS;
      e.id = ~0ull;
      oc[g] = e;
    }
  }
  // With an owner output (gathered shard lists) `pos` travels through the selector as the candidate's slot index t
  // (source = t / kk_in) and the record's own position is fetched when the row is emitted; otherwise the position
  // itself travels: no dependent load per emitted row (ten serial round trips in a single query's merge)
  const bool by_slot = a.out_owner != nullptr;
  auto emit_row = [&](uint32_t rk, float d, uint32_t t, uint64_t id) {
    uint32_t sidx = 0, pos = t;
    if (by_slot) {
      sidx = t / a.kk_in;
      pos = src[(size_t)sidx * a.src_stride + t % a.kk_in].pos;
    }
    if (oi) oi[rk] = id;
    if (od) od[rk] = d;
    if (op) op[rk] = pos;
    if (oo) oo[rk] = sidx;
    if (oc) {
      Cand e;
      e.d = d;
      e.pos = pos;
      e.id = id;
      oc[rk] = e;
    }
  };
  __shared__ uint32_t s_pre[MERGE_PRE_SRC + 1];
  if constexpr (NW > 1) {
    __shared__ uint32_t s_pfx[MERGE_PRE_SRC + 1], s_wt[NW];
    constexpr uint32_t NT = MI355_WAVE * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char merge_lds[];
    float* sl_d = (float*)merge_lds;                      // [MERGE_SHORT_CAP] (the sample's keys before the sweep)
    uint32_t* sl_t = (uint32_t*)(sl_d + MERGE_SHORT_CAP);  // [MERGE_SHORT_CAP] slot index
    uint32_t* fl = sl_t + MERGE_SHORT_CAP;                // [MERGE_FINAL_CAP] indices into the short list
    uint64_t* f_id = (uint64_t*)(fl + MERGE_FINAL_CAP);   // [MERGE_FINAL_CAP] the final list: row ids,
    float* f_d = (float*)(f_id + MERGE_FINAL_CAP);        // [MERGE_FINAL_CAP] distances (-0 as +0),
    uint32_t* f_t = (uint32_t*)(f_d + MERGE_FINAL_CAP);   // [MERGE_FINAL_CAP] slots
    uint32_t* hist = f_t + MERGE_FINAL_CAP;               // [256]
    uint32_t* st = hist + 256;                            // [0..1] radix state, [2] short-list rows, [3] final rows
    if (a.src_cnt && a.n_src <= MERGE_PRE_SRC && a.k_out <= NT) {  // block-uniform
      uint32_t filled = 0;
      for (uint32_t sidx = tid; sidx < a.n_src; sidx += NT) {
        const uint32_t c = min(a.src_cnt[(size_t)sidx * a.cnt_stride + (size_t)b * a.cnt_q_stride], a.kk_in);
        s_pre[sidx] = c;
        filled += c;
      }
      if (tid == 0) st[2] = st[3] = st[4] = 0;
      __syncthreads();
      if (filled) atomicAdd(&st[4], filled);
      __syncthreads();
      const uint32_t n_filled = st[4];
      auto key_of = [&](const Cand& c) -> uint32_t {  // 0xFFFFFFFF: not a candidate; -0 orders as +0, like the compare
        return (c.pos != CAND_EMPTY_POS && c.d == c.d) ? f32_sort_key(c.d) : 0xFFFFFFFFu;
      };
      // A query whose filled slots fit one row per thread (the items of a sliced batch share their bounds: a single query
      // leaves a few dozen rows in ~240 of its 1024 sources): every thread fetches ONE filled slot — found by a binary search
      // over the prefix of the counts — and the rows are ranked against each other at once.  No sweep over the empty slots, no
      // radix passes, no second fetch of the winners: 12.5 -> ~7 us for a single query at C3.
      if (n_filled <= NT && a.n_src <= NT) {  // block-uniform
        {
          const uint32_t c = (uint32_t)tid < a.n_src ? s_pre[tid] : 0u;
          uint32_t inc = c;
#pragma unroll
          for (int off = 1; off < MI355_WAVE; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
          }
          if (lane == MI355_WAVE - 1) s_wt[wid] = inc;
          __syncthreads();
          uint32_t base = 0;
          for (int w = 0; w < wid; ++w) base += s_wt[w];
          if ((uint32_t)tid < a.n_src) s_pfx[tid] = base + inc - c;
          if (tid == 0) s_pfx[a.n_src] = n_filled;
        }
        __syncthreads();
        Cand mine;
        mine.d = 0.f;
        mine.pos = CAND_EMPTY_POS;
        mine.id = 0;
        uint32_t my_t = 0;
        bool valid = false;
        if ((uint32_t)tid < n_filled) {
          uint32_t lo = 0, hi = a.n_src;  // the last source whose prefix is <= tid holds filled slot tid
          while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_pfx[mid] <= (uint32_t)tid) lo = mid; else hi = mid;
          }
          const uint32_t r = (uint32_t)tid - s_pfx[lo];
          mine = src[(size_t)lo * a.src_stride + r];
          my_t = lo * a.kk_in + r;
          valid = key_of(mine) != 0xFFFFFFFFu;
        }
        // every wave keeps its own k_out best rows (a radix select on ballots, one key per lane): at most 16 * k_out rows are
        // ranked against each other — ranking all filled rows is quadratic on ONE CU (800 rows: 30 us, round 6)
        const uint32_t mykey = valid ? f32_sort_key(mine.d) : 0xFFFFFFFFu;
        uint32_t t_w = 0xFFFFFFFFu;
        if (a.k_out <= (uint32_t)MI355_WAVE) t_w = wave_kth_smallest_key(mykey, a.k_out);  // (fewer valid rows: 0xFFFFFFFF, keeps all)
        const bool keep = valid && mykey <= t_w;
        uint32_t my_i = 0;
        {
          const uint64_t mv = __ballot(valid), mk = __ballot(keep);
          if (lane == 0 && mv) atomicAdd(&st[3], (uint32_t)__popcll((unsigned long long)mv));
          uint32_t base = 0;
          if (lane == 0 && mk) base = atomicAdd(&st[2], (uint32_t)__popcll((unsigned long long)mk));
          base = __shfl(base, 0);
          my_i = base + (uint32_t)__popcll((unsigned long long)(mk & ((1ull << lane) - 1ull)));
        }
        if (keep) {
          f_d[my_i] = mine.d + 0.0f;
          f_id[my_i] = mine.id;
        }
        __syncthreads();
        const uint32_t n_surv = st[2];
        if (keep) {
          const float d = mine.d + 0.0f;
          uint32_t rank = 0;
          for (uint32_t j0 = 0; j0 < n_surv; j0 += 4) {  // broadcast reads, four distances at a time
            const float4 dj = *(const float4*)&f_d[j0];
            const float dv[4] = {dj.x, dj.y, dj.z, dj.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (j0 + e >= n_surv) break;
              bool lt = dv[e] < d;
              if (dv[e] == d) {
                const uint64_t idj = f_id[j0 + e];
                lt = idj < mine.id || (idj == mine.id && j0 + e < my_i);
              }
              rank += lt ? 1u : 0u;
            }
          }
          if (rank < a.k_out) emit_row(rank, mine.d, by_slot ? my_t : mine.pos, mine.id);
        }
        if (tid == 0) a.out_cnt[b] = min(st[3], a.k_out);
#ifdef MI355_DEV_COUNTERS  // dev[5]: 7e8 + filled slots = this path
        if (tid == 0 && a.ctl) atomicAdd(const_cast<uint32_t*>(&a.ctl->dev[5]), 700000000u + n_filled);
#endif
        return;
      }
      uint32_t tau = 0xFFFFFFFFu;
      if (n_filled > MERGE_SHORT_CAP) {
        // Two samples of NT distinct filled slots, the smaller k_out-th key wins.  (A) the first rows of every source:
        // tight when the sources are long sorted lists whose heads hold the winners — but short of k_out keys when
        // most work items returned nothing (a bound shared early); (B) every (n_filled / NT)-th filled slot: its
        // k_out-th key sits near rank k_out * n_filled / NT of all slots — enough exactly when the lists are short.
        uint32_t* s_keys = (uint32_t*)sl_d;
        {
          const uint32_t sidx = (uint32_t)tid % a.n_src, r = (uint32_t)tid / a.n_src;
          uint32_t key = 0xFFFFFFFFu;
          if (r < s_pre[sidx]) key = key_of(src[(size_t)sidx * a.src_stride + r]);
          s_keys[tid] = key;
        }
        // exclusive prefix of the counts (n_src <= NT: one source per thread)
        {
          const uint32_t c = (uint32_t)tid < a.n_src ? s_pre[tid] : 0u;
          uint32_t inc = c;
#pragma unroll
          for (int off = 1; off < MI355_WAVE; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
          }
          if (lane == MI355_WAVE - 1) s_wt[wid] = inc;
          __syncthreads();
          uint32_t base = 0;
          for (int w = 0; w < wid; ++w) base += s_wt[w];
          if ((uint32_t)tid < a.n_src) s_pfx[tid] = base + inc - c;
          if (tid == 0) s_pfx[a.n_src] = n_filled;
        }
        __syncthreads();
        const uint32_t tau_a = block_kth_smallest_key<NT>([&](uint32_t i) { return s_keys[i]; }, NT, a.k_out, hist, st);
        __syncthreads();
        {
          const uint32_t f = (uint32_t)(((uint64_t)tid * n_filled) / NT);  // distinct: n_filled > NT
          uint32_t lo = 0, hi = a.n_src;  // the last source whose prefix is <= f holds filled slot f
          while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_pfx[mid] <= f) lo = mid; else hi = mid;
          }
          s_keys[tid] = key_of(src[(size_t)lo * a.src_stride + (f - s_pfx[lo])]);
        }
        __syncthreads();
        const uint32_t tau_b = block_kth_smallest_key<NT>([&](uint32_t i) { return s_keys[i]; }, NT, a.k_out, hist, st);
        tau = min(tau_a, tau_b);
        __syncthreads();  // the keys are dead: the sweep reuses their space
      }
      constexpr int G = 8;  // (a single query cut by rows has up to 1024 sources: 10 240 slots at k = 10, two trips)
      for (uint32_t t0 = 0; t0 < n; t0 += G * NT) {
        Cand c[G];
        bool in[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
          // (unconditional load at a clamped slot: guarded, hipcc waits for each of the G records on its own — the
          //  eight serial round trips were most of a single query's 12 us merge)
          const uint32_t t = t0 + (uint32_t)u * NT + tid, tc = t < n ? t : n - 1u;
          in[u] = t < n && t % a.kk_in < s_pre[tc / a.kk_in];
          c[u] = src[(size_t)(tc / a.kk_in) * a.src_stride + tc % a.kk_in];
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
          const uint32_t t = t0 + (uint32_t)u * NT + tid;
          const uint32_t key = in[u] ? key_of(c[u]) : 0xFFFFFFFFu;
          const bool take = key != 0xFFFFFFFFu && key <= tau;
          const uint64_t m = __ballot(take);
          if (m) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&st[2], (uint32_t)__popcll((unsigned long long)m));
            base = __shfl(base, 0);
            const uint32_t at = base + (uint32_t)__popcll((unsigned long long)(m & ((1ull << lane) - 1ull)));
            if (take && at < MERGE_SHORT_CAP) {
              sl_d[at] = c[u].d + 0.0f;
              sl_t[at] = t;
            }
          }
        }
      }
      __syncthreads();
      const uint32_t n_short = st[2];
      bool done = false;
      if (n_short <= MERGE_SHORT_CAP) {
        uint32_t tau2 = 0xFFFFFFFFu;
        if (n_short > a.k_out)
          tau2 = block_kth_smallest_key<NT>([&](uint32_t i) { return f32_sort_key(sl_d[i]); }, n_short, a.k_out, hist, st);
        for (uint32_t i0 = 0; i0 < n_short; i0 += NT) {
          const uint32_t i = i0 + tid;
          const bool take = i < n_short && f32_sort_key(sl_d[i]) <= tau2;
          const uint64_t m = __ballot(take);
          if (m) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&st[3], (uint32_t)__popcll((unsigned long long)m));
            base = __shfl(base, 0);
            const uint32_t at = base + (uint32_t)__popcll((unsigned long long)(m & ((1ull << lane) - 1ull)));
            if (take && at < MERGE_FINAL_CAP) fl[at] = i;
          }
        }
        __syncthreads();
        const uint32_t n_fin = st[3];
#ifdef MI355_DEV_COUNTERS  // dev[5] (the scan's redone passes, 0 up to kk 128) doubles as: short rows + 1e5 * final rows
        if (tid == 0 && a.ctl) atomicAdd(const_cast<uint32_t*>(&a.ctl->dev[5]), n_short + 100000u * min(n_fin, 9999u));
#endif
        if (n_fin <= MERGE_FINAL_CAP) {
          for (uint32_t i = tid; i < n_fin; i += NT) {
            const uint32_t me = fl[i], t = sl_t[me];
            f_d[i] = sl_d[me];
            f_t[i] = t;
            f_id[i] = src[(size_t)(t / a.kk_in) * a.src_stride + t % a.kk_in].id;
          }
          __syncthreads();
          for (uint32_t i = tid; i < n_fin; i += NT) {
            const float d = f_d[i];
            const uint64_t id = f_id[i];
            uint32_t rank = 0;
            for (uint32_t j0 = 0; j0 < n_fin; j0 += 4) {  // broadcast reads, four distances at a time
              const float4 dj = *(const float4*)&f_d[j0];
              const float dv[4] = {dj.x, dj.y, dj.z, dj.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if (j0 + e >= n_fin) break;
                bool lt = dv[e] < d;
                if (dv[e] == d) {  // a tie on the distance: the row ids decide; the same row twice: its place in the list
                  const uint64_t idj = f_id[j0 + e];
                  lt = idj < id || (idj == id && j0 + e < i);
                }
                rank += lt ? 1u : 0u;
              }
            }
            if (rank < a.k_out) {
              const uint32_t t = f_t[i];
              const Cand c = src[(size_t)(t / a.kk_in) * a.src_stride + t % a.kk_in];
              emit_row(rank, c.d, by_slot ? t : c.pos, c.id);
            }
          }
          if (tid == 0) a.out_cnt[b] = min(n_fin, a.k_out);
          done = true;
        }
      }
#ifdef MI355_DEV_COUNTERS
      if (!done && tid == 0 && a.ctl) atomicAdd(const_cast<uint32_t*>(&a.ctl->dev[5]), 1000000000u);
#endif
      if (done) return;
    }
    if (wid != 0) return;
  }
  // With counts and up to MERGE_PRE_SRC sources the filled slots are addressed directly: the counts are loaded
  // together (one global round trip), their exclusive prefix goes to LDS, and filled slot t of the query is found
  // by a binary search over it — a scan leaves a handful of rows in most work items once the query has a bound, so
  // a walk over the filled slots only costs two round trips plus a tile of 64 per 64 rows actually there.
  const bool compact = a.src_cnt && a.n_src <= MERGE_PRE_SRC;
  uint32_t n_filled = 0;
  if (compact) {
    constexpr int G = 8;
    for (uint32_t s0 = 0; s0 < a.n_src; s0 += G * MI355_WAVE) {
      uint32_t c[G];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const uint32_t sidx = s0 + u * MI355_WAVE + lane;
        c[u] = sidx < a.n_src ? min(a.src_cnt[(size_t)sidx * a.cnt_stride + (size_t)b * a.cnt_q_stride], a.kk_in) : 0u;
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (s0 + u * MI355_WAVE >= a.n_src) break;
        uint32_t inc = c[u];
#pragma unroll
        for (int off = 1; off < MI355_WAVE; off <<= 1) {
          const uint32_t o = __shfl_up(inc, off);
          if (lane >= off) inc += o;
        }
        const uint32_t sidx = s0 + u * MI355_WAVE + lane;
        if (sidx < a.n_src) s_pre[sidx] = n_filled + inc - c[u];
        n_filled += __shfl(inc, MI355_WAVE - 1);
      }
    }
    if (lane == 0) s_pre[a.n_src] = n_filled;
    __syncthreads();
  }
  auto gen = [&](WaveTopK<KPL>& top) {
    if (compact) {
      constexpr int G = 8;
      for (uint32_t t0 = 0; t0 < n_filled; t0 += G * MI355_WAVE) {
        Cand c[G];
        bool ok[G];
        uint32_t slot[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
          const uint32_t t = t0 + u * MI355_WAVE + lane;
          c[u].d = 0.f;
          c[u].pos = CAND_EMPTY_POS;
          c[u].id = 0;
          ok[u] = false;
          slot[u] = 0;
          if (t0 + u * MI355_WAVE >= n_filled) break;
          uint32_t lo = 0, hi = a.n_src;  // the last source whose prefix is <= t holds slot t (empty sources share a prefix)
          while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_pre[mid] <= t) lo = mid; else hi = mid;
          }
          if (t < n_filled) {
            const uint32_t r = t - s_pre[lo];
            slot[u] = lo * a.kk_in + r;
            c[u] = src[(size_t)lo * a.src_stride + r];
            ok[u] = c[u].pos != CAND_EMPTY_POS && c[u].d == c[u].d;
          }
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
          if (t0 + u * MI355_WAVE < n_filled) top.offer(ok[u], c[u].d, by_slot ? slot[u] : c[u].pos, c[u].id, lane);
        }
      }
      return;
    }
    if (a.src_cnt && a.kk_in > 32u) {
      // long lists with counts: walk source by source and read only the filled part (with a query bound in
      // place most of a scan's work items return a handful of rows, or none, in their kk slots)
      for (uint32_t sidx = 0; sidx < a.n_src; ++sidx) {
        const uint32_t lim = min(a.src_cnt[(size_t)sidx * a.cnt_stride + (size_t)b * a.cnt_q_stride], a.kk_in);
        for (uint32_t i0 = 0; i0 < lim; i0 += MI355_WAVE) {
          const uint32_t i = i0 + lane;
          Cand c;
          c.d = 0.f;
          c.pos = CAND_EMPTY_POS;
          c.id = 0;
          bool ok = false;
          if (i < lim) {
            c = src[(size_t)sidx * a.src_stride + i];
            ok = c.pos != CAND_EMPTY_POS && c.d == c.d;
          }
          top.offer(ok, c.d, by_slot ? sidx * a.kk_in + i : c.pos, c.id, lane);
        }
      }
      return;
    }
    // eight steps of 64 slots at a time: their count loads, then their record loads, are all in flight
    // together (a step-by-step loop is two dependent global round trips per 64 slots: 60 us for the 2560
    // slots of one sliced single query)
    constexpr int G = 8;
    for (uint32_t t0 = 0; t0 < n; t0 += G * MI355_WAVE) {
      uint32_t lim[G];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const uint32_t t = t0 + u * MI355_WAVE + lane;
        lim[u] = 0;
        if (t < n) lim[u] = a.src_cnt ? min(a.src_cnt[(size_t)(t / a.kk_in) * a.cnt_stride + (size_t)b * a.cnt_q_stride], a.kk_in) : a.kk_in;
      }
      Cand c[G];
      bool ok[G];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const uint32_t t = t0 + u * MI355_WAVE + lane;
        c[u].d = 0.f;
        c[u].pos = CAND_EMPTY_POS;
        c[u].id = 0;
        ok[u] = false;
        if (t < n && t % a.kk_in < lim[u]) {
          c[u] = src[(size_t)(t / a.kk_in) * a.src_stride + t % a.kk_in];
          ok[u] = c[u].pos != CAND_EMPTY_POS && c[u].d == c[u].d;
        }
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (t0 + u * MI355_WAVE < n) top.offer(ok[u], c[u].d, by_slot ? t0 + u * MI355_WAVE + lane : c[u].pos, c[u].id, lane);
      }
    }
  };
  const uint32_t n_out = wave_select_sorted<KPL>(a.k_out, lane, gen, emit_row);
  if (lane == 0) a.out_cnt[b] = n_out;
}

// mi355_merge_topk: n_lists lists of k per query, arrays-of-fields layout
template <int KPL>
__global__ __launch_bounds__(64) void k_merge_lists(const uint64_t* __restrict__ in_ids,
                                                    const float* __restrict__ in_dist,
                                                    const uint32_t* __restrict__ in_cnt,
                                                    uint32_t n_lists, uint32_t nq, uint32_t k,
                                                    uint64_t* __restrict__ out_ids,
                                                    float* __restrict__ out_dist,
                                                    uint32_t* __restrict__ out_cnt) {
  const int lane = threadIdx.x;
  const uint32_t b = blockIdx.x;
  const uint32_t n = n_lists * k;
  uint64_t* oi = out_ids + (size_t)b * k;
  float* od = out_dist + (size_t)b * k;
  for (uint32_t g = lane; g < k; g += MI355_WAVE) {
    oi[g] = ~0ull;
    od[g] = __builtin_huge_valf();
  }
  auto gen = [&](WaveTopK<KPL>& top) {
    for (uint32_t t0 = 0; t0 < n; t0 += MI355_WAVE) {
      uint32_t t = t0 + lane;
      bool ok = false;
      float d = 0.f;
      uint64_t id = 0;
      if (t < n) {
        uint32_t l = t / k, i = t % k;
        uint32_t cnt = min(in_cnt[(size_t)l * nq + b], k);
        if (i < cnt) {
          size_t o = ((size_t)l * nq + b) * k + i;
          d = in_dist[o];
          id = in_ids[o];
          ok = d == d;
        }
      }
      top.offer(ok, d, 0u, id, lane);
    }
  };
  const uint32_t n_out = wave_select_sorted<KPL>(k, lane, gen, [&](uint32_t rk, float d, uint32_t, uint64_t id) {
    oi[rk] = id;
    od[rk] = d;
  });
  if (lane == 0) out_cnt[b] = n_out;
}

// ------------------------------------------------------------------ K6 -----
// exact distance of the flat / refine path on one raw row: the contract's
// d-ascending chains.  The ORDER is sequential; the LOADS are not: rows are read
// in 16-B pieces, 4 pieces (64 B) in flight per lane, because one thread per row
// with scalar loads is latency-bound (measured: 3.1 ms of the 23.7 ms C2 step).
struct DistAcc {
  float l2, qv, vv;
};
__device__ __forceinline__ void dist_step(DistAcc& a, float qd, float v, uint32_t metric) {
  if (metric == MI355_METRIC_L2) {
    const float t = qd - v;
    a.l2 = __fmaf_rn(t, t, a.l2);
  } else {
    a.qv = __fmaf_rn(qd, v, a.qv);
    a.vv = __fmaf_rn(v, v, a.vv);
  }
}
__device__ __forceinline__ float dist_finish(const DistAcc& a, uint32_t metric, float qq) {
  if (metric == MI355_METRIC_L2) return a.l2;
  if (metric == MI355_METRIC_DOT) return 1.0f - a.qv;
  return 1.0f - ieee_divf(a.qv, ieee_sqrtf(qq) * ieee_sqrtf(a.vv));
}

#ifndef MI355_REFINE_SIDE_PIECES
#define MI355_REFINE_SIDE_PIECES 8  // 16-B pieces in flight per lane of the re-rank that runs beside a scan (rows arrive over PCIe)
#endif
template <int PIECES = 4>  // 16-B pieces of the row in flight per lane
__device__ __forceinline__ float exact_distance(const float* __restrict__ q, const void* raw,
                                                uint32_t dtype, uint64_t row, uint32_t dim,
                                                uint32_t metric, float qq) {
  const uint64_t base = row * dim;
  DistAcc a = {0.f, 0.f, 0.f};
  const uint32_t esz = dtype == MI355_DTYPE_F32 ? 4u : 2u;
  const uint32_t per = 16u / esz;  // elements per 16-B piece
  const unsigned char* p = (const unsigned char*)raw + base * esz;
  if ((dim % per) == 0 && (((size_t)p) & 15u) == 0) {
    const uint32_t n_pieces = dim / per;
    // (ext-vector registers, unconditional loads at a clamped piece: an array of more than eight HIP uint4 structs lives in
    //  scratch and every load into it is waited for on its own — scripts/check_scratch.py)
    typedef __attribute__((ext_vector_type(4))) uint32_t ed_u32x4;
    const ed_u32x4* pv = (const ed_u32x4*)p;
    uint32_t d = 0;
    for (uint32_t i0 = 0; i0 < n_pieces; i0 += PIECES) {
      ed_u32x4 buf[PIECES];
#pragma unroll
      for (int u = 0; u < PIECES; ++u) buf[u] = pv[min(i0 + (uint32_t)u, n_pieces - 1u)];
#pragma unroll
      for (int u = 0; u < PIECES; ++u) {
        if (i0 + u >= n_pieces) continue;  // (not `break`: an early exit keeps the loop rolled and buf[] in scratch)
        const uint32_t w[4] = {buf[u].x, buf[u].y, buf[u].z, buf[u].w};
        if (dtype == MI355_DTYPE_F32) {
#pragma unroll
          for (int e = 0; e < 4; ++e) dist_step(a, q[d + e], __uint_as_float(w[e]), metric);
          d += 4;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint16_t lo = (uint16_t)(w[e] & 0xffffu), hi = (uint16_t)(w[e] >> 16);
            const float v0 = dtype == MI355_DTYPE_BF16 ? bf16_bits_to_f32(lo) : f16_bits_to_f32(lo);
            const float v1 = dtype == MI355_DTYPE_BF16 ? bf16_bits_to_f32(hi) : f16_bits_to_f32(hi);
            dist_step(a, q[d + 2 * e], v0, metric);
            dist_step(a, q[d + 2 * e + 1], v1, metric);
          }
          d += 8;
        }
      }
    }
  } else {
    for (uint32_t d = 0; d < dim; ++d) dist_step(a, q[d], load_elem(raw, dtype, base + d), metric);
  }
  return dist_finish(a, metric, qq);
}

// Refine (query.rs:1313-1317): exact distance for the kk approximate winners, range filter
// on the exact distance; the (distance, rowid) sort + keep k is k_merge_cands over the
// records written here (any kk).  grid = (ceil(kk / 256), nq); a thread per candidate.
// `owner` (sharded search): only candidates whose raw vector lives on this rank are refined,
// the others stay empty and arrive through the second all-gather from their owner.
struct RefineArgs {
  IndexView ix;
  const float* q;          // ORIGINAL queries [nq, dim]
  const Cand* in;          // [nq, kk] ANN winners (pos = local row position on the OWNING handle)
  const uint32_t* in_cnt;  // [nq]
  const uint32_t* in_owner;  // [nq, kk] or nullptr (everything is local)
  uint32_t my_rank;
  uint32_t kk;
  RangeFilter range;
  Cand* out;               // [nq, kk]
  const DevCtl* ctl;
  uint32_t n_rows;         // rows on this handle: a position at or past it is not refined (a peer's slab
                           // that a timed-out scan left unwritten must not become an address)
  uint32_t nq;             // queries of this launch (grid.y may be smaller: the workgroups stride over them)
  uint32_t side_slots;     // SIDE kernel: threads per query, a multiple of 64 that divides 256 (256 when kk > 128)
  ActiveMask act;
};

// (Round 5 measured a coalesced variant of the host-column re-rank — eight lanes per 128 contiguous bytes of a row, rows
//  staged through LDS, one chain per lane — against this one-row-per-lane kernel: 23.6 against 29.5 GB/s at a 61 GB
//  column, 16.5 against 20.6 GB/s at 200 GB.  The rate falls with the column's footprint, not with the request shape:
//  the gather is bound by address translation of random 3-KiB rows in host memory; profiles/r05_c5_gather_ab.txt.)
// SIDE = false: the query and its |q|^2 live in LDS (one workgroup per query).
// SIDE = true (the deferred re-rank that runs BESIDE the next call's scan on a few reserved CUs): no LDS at
// all and a small grid striding over the queries — the scan's workgroups take a whole CU each (128 VGPRs x 16
// waves, 159 KiB of LDS), so whatever overlaps them must fit on the CUs they leave free; the query is read with
// wave-uniform (scalar) loads, |q|^2 is one lane's chain broadcast by readlane, eight row pieces in flight per lane.
template <bool SIDE>
static __global__ __launch_bounds__(256) void k_refine_dist(RefineArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float s_qq_lds[SIDE ? 1 : 2];
  const int tid = threadIdx.x;
  if (a.ctl && a.ctl->timed_out) return;
  // SIDE: a workgroup takes 256 / slots queries at a time (slots = kk rounded up to whole waves), so that all its
  // threads keep row pieces in flight; otherwise one query per workgroup (grid.x covers kk)
  const uint32_t slots = SIDE ? a.side_slots : 256u, qpb = 256u / slots;
  for (uint32_t b0 = blockIdx.y * qpb; b0 < a.nq; b0 += gridDim.y * qpb) {
    const uint32_t b = b0 + (SIDE ? (uint32_t)tid / slots : 0u);
    if (b >= a.nq) continue;  // (SIDE has no barrier below)
    if (!a.act.on(b)) continue;
    const float* q = a.q + (size_t)b * a.ix.dim;
    const float* sq = q;
    float qq;
    if constexpr (!SIDE) {
      float* l = (float*)smem;  // [dim]
      for (uint32_t d = tid; d < a.ix.dim; d += 256) l[d] = q[d];
      __syncthreads();
      if (tid == 0) {
        float acc = 0.f;
        for (uint32_t d = 0; d < a.ix.dim; ++d) acc = __fmaf_rn(l[d], l[d], acc);
        s_qq_lds[0] = acc;
      }
      __syncthreads();
      sq = l;
      qq = s_qq_lds[0];
    } else {
      float acc = 0.f;
      if ((tid & 63) == 0)
        for (uint32_t d = 0; d < a.ix.dim; ++d) acc = __fmaf_rn(q[d], q[d], acc);
      qq = readlane_f(acc, 0);
    }
    const uint32_t cnt = min(a.in_cnt[b], a.kk);
    const uint32_t c = SIDE ? blockIdx.x * 256 + (uint32_t)tid % slots : blockIdx.x * 256 + tid;
    if (c < a.kk) {
      Cand o;
      o.d = __builtin_huge_valf();
      o.pos = CAND_EMPTY_POS;
      o.id = ~0ull;
      if (c < cnt && (!a.in_owner || a.in_owner[(size_t)b * a.kk + c] == a.my_rank)) {
        const Cand in = a.in[(size_t)b * a.kk + c];
        if (in.pos != CAND_EMPTY_POS && in.pos < a.n_rows) {
          const uint64_t rrow = a.ix.raw_by_global ? global_pos_of(a.ix, in.pos) : (uint64_t)in.pos;
          const float d = exact_distance<SIDE ? MI355_REFINE_SIDE_PIECES : 4>(sq, a.ix.raw, a.ix.raw_dtype, rrow, a.ix.dim, a.ix.metric, qq);
          if (in_range(d, a.range)) {
            o.d = d;
            o.pos = in.pos;
            o.id = in.id;
          }
        }
      }
      a.out[(size_t)b * a.kk + c] = o;
    }
    if constexpr (!SIDE) __syncthreads();  // the LDS copy of the query is rebuilt for the next one
  }
}

// arrays-of-fields result lists ([nq, k] ids / distances + [nq] counts) -> packed Cand records
// (the record the sharded search all-gathers); pos carries the slot index
static __global__ void k_pack_cands(const uint64_t* __restrict__ ids, const float* __restrict__ dist,
                                    const uint32_t* __restrict__ cnt, uint32_t nq, uint32_t k, Cand* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq * k) return;
  const uint32_t b = i / k, r = i % k;
  Cand c;
  c.d = __builtin_huge_valf();
  c.pos = CAND_EMPTY_POS;
  c.id = ~0ull;
  if (r < min(cnt[b], k)) {
    c.d = dist[i];
    c.pos = r;
    c.id = ids[i];
  }
  out[i] = c;
}

// maximum_nprobes expansion: the short queries' vectors gathered into a dense batch, and their
// second-pass results written back over the first pass's rows
// the queries whose ANN stage found fewer than kk rows, in ascending query order: rows[0 .. *n_short)
// (one 1024-thread block; the count stays on the device, see ActiveMask) and the counter of the call
static __global__ __launch_bounds__(1024) void k_compact_short(const uint32_t* __restrict__ cnt_ann, uint32_t nq, uint32_t kk,
                                                               uint32_t* __restrict__ rows, uint32_t* __restrict__ n_short,
                                                               DevCtl* __restrict__ ctl) {
  __shared__ uint32_t s_wave[16], s_base;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (uint32_t q0 = 0; q0 < nq; q0 += 1024u) {
    const uint32_t q = q0 + tid;
    const bool is_short = q < nq && cnt_ann[q] < kk;
    const unsigned long long bal = __ballot(is_short);
    if (lane == 0) s_wave[wid] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t base = s_base;
    for (uint32_t w = 0; w < wid; ++w) base += s_wave[w];
    if (is_short) rows[base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = q;
    __syncthreads();
    if (tid == 0) {
      uint32_t t = s_base;
      for (uint32_t w = 0; w < 16; ++w) t += s_wave[w];
      s_base = t;
    }
    __syncthreads();
  }
  if (tid == 0) {
    *n_short = s_base;
    if (ctl) atomicAdd(&ctl->short_queries, s_base);
  }
}

static __global__ __launch_bounds__(256) void k_gather_rows_f32(const float* __restrict__ src, const uint32_t* __restrict__ rows,
                                                              uint32_t dim, float* __restrict__ dst,
                                                              ActiveMask act = ActiveMask()) {
  if (!act.on(blockIdx.x)) return;
  const float* s = src + (size_t)rows[blockIdx.x] * dim;
  float* d = dst + (size_t)blockIdx.x * dim;
  for (uint32_t i = threadIdx.x; i < dim; i += 256) d[i] = s[i];
}

static __global__ __launch_bounds__(64) void k_scatter_results(const uint32_t* __restrict__ rows, uint32_t k,
                                                              const uint64_t* __restrict__ s_ids, const float* __restrict__ s_dist,
                                                              const uint32_t* __restrict__ s_cnt, uint64_t* __restrict__ ids,
                                                              float* __restrict__ dist, uint32_t* __restrict__ cnt,
                                                              ActiveMask act = ActiveMask()) {
  if (!act.on(blockIdx.x)) return;
  const uint32_t i = blockIdx.x, q = rows[i];
  for (uint32_t g = threadIdx.x; g < k; g += 64) {
    ids[(size_t)q * k + g] = s_ids[(size_t)i * k + g];
    dist[(size_t)q * k + g] = s_dist[(size_t)i * k + g];
  }
  if (threadIdx.x == 0) cnt[q] = s_cnt[i];
}

// ------------------------------------------------------- index packing -----
// Re-pack one partition's codes into the device layout [m][pstride] (zero
// padded).  src is either the caller's row-major rows of this partition
// ([len, m]) or lance's transposed block ([m, len]).  grid = (tiles of 64 rows,
// partitions in this batch).
struct RepackArgs {
  const uint8_t* src;         // base of the staged chunk
  const uint64_t* src_off;    // [n_parts] byte offset of each partition inside src
  const uint32_t* part_ids;   // [n_parts] partition ids
  uint8_t* dst;
  const uint64_t* code_off;   // [nlist]
  const uint32_t* plen;
  const uint32_t* pstride;
  uint32_t m;
  uint32_t transposed;
};

static __global__ __launch_bounds__(256) void k_repack_codes(RepackArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tile[];  // [64][m+1]
  const uint32_t p = a.part_ids[blockIdx.y];
  const uint32_t len = a.plen[p], stride = a.pstride[p];
  const uint32_t r0 = blockIdx.x * 64;
  if (r0 >= stride) return;
  const uint8_t* src = a.src + a.src_off[blockIdx.y];
  uint8_t* dst = a.dst + a.code_off[p];
  const uint32_t m = a.m, pitch = m + 1;
  const uint32_t nrow = min(64u, stride - r0);
  for (uint32_t e = threadIdx.x; e < 64u * m; e += 256) {
    uint32_t v = 0;
    if (a.transposed) {
      uint32_t j = e / 64u, i = e % 64u;
      if (r0 + i < len) v = src[(size_t)j * len + r0 + i];
      tile[i * pitch + j] = (uint8_t)v;
    } else {
      uint32_t i = e / m, j = e % m;
      if (r0 + i < len) v = src[(size_t)(r0 + i) * m + j];
      tile[i * pitch + j] = (uint8_t)v;
    }
  }
  __syncthreads();
  for (uint32_t e = threadIdx.x; e < 64u * m; e += 256) {
    uint32_t j = e / 64u, i = e % 64u;
    if (i < nrow) dst[(size_t)j * stride + r0 + i] = tile[i * pitch + j];
  }
}
