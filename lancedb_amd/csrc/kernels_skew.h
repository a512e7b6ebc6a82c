// kernels_skew.h — the production ADC scan: bank-conflict-free LUT gathers.
//
// Why.  The ADC inner loop (lance-index pq distance:
// dist[i] = sum_j LUT[j][code[i][j]], j ascending, plain f32 adds) costs one
// 4-byte LDS gather per code byte at a data-dependent address.  With the table
// stored [j][code] all 64 lanes of a wave are at the same j and random codes
// collide on the 32 banks of ds_read_b32 (~3.4-way on average): the kernel is
// LDS-bound at a third of the LDS rate (measured: profiles/r01_a_*).
//
// How.  Store the table [code][column] with a pitch of P dwords, P % 32 == 0,
// and let the lanes of each 32-lane bank group run at 32 different phases of
// their rows: lane l (lm = l & 31) is lm steps behind lane 0, so at step t it
// needs column j = t - lm of ITS code: bank = (code*P + j) % 32 = (t - lm) % 32,
// distinct for the 32 lanes of a group whatever the codes are.  Every lane
// still adds its row's LUT values in j-ascending order, so results stay
// bit-identical to the oracle's chain.
//
//  * Time skew across rows.  A wave owns a "stream" of 64-row tiles; lane l
//    holds row l of each tile.  In the M steps of tile n, lanes with lm > t are
//    still finishing row n-1 (j = M + t - lm).  Two accumulators (X = row n,
//    Y = row n-1) are selected by a compile-time EXEC mask in the first 31
//    steps of each tile; after step 30 Y is complete for all 64 lanes and is
//    offered to the top-k selection in one go.
//  * Affine addresses.  Columns are indexed u = t + 32 - lm in [1, M+31]; the
//    LUT columns j >= M-31 are stored twice (u = j+32 and u = j+32-M) so that
//    address = code*P*4 + 4*(32-lm) + 4*t holds for both rows in flight:
//    one shift-add per gather, 4*t in the ds_read offset field.
//  * Pre-skewed storage.  The per-lane byte streams are laid out at index-open
//    time in exactly the order the lanes consume them: 1 KiB chunks
//    [64 lanes][16 steps], i.e. one global_load_dwordx4 per lane per 16 steps,
//    fully coalesced.  A partition is split into SK_STREAMS streams (tile g
//    belongs to stream g % 16) so that the 16 waves of a workgroup are balanced
//    to within one tile.  Each stream carries 32 extra steps (2 chunks) for the
//    tails of its last tile: +1.4 % bytes at C3's partition sizes.
//
// Work distribution.  One work item = one (query, probed partition) pair, as in
// the reference's ANNIvfSubIndexExec (table/query.rs:1079), but the items of a
// batch are sorted by partition and queued per XCD: the 32 CUs of an XCD pull
// consecutive items, i.e. the SAME partition for different queries, so the
// partition's codes are read from HBM once and then served by that XCD's 4 MiB
// L2.  Placement only affects speed; any CU may steal from any queue.
#pragma once
#include "kernels_ivfpq.h"

#define SK_STREAMS 16u
#define SK_TILE 64u
#define SK_TAIL_CHUNKS 2u  // 32 skew steps / 16 steps per chunk
#define SK_NONE 0xFFFFFFFFu
#define SK_HEAD_STRIDE 32u  // u32 words between the per-XCD queue heads (128 B)

__host__ __device__ __forceinline__ uint32_t sk_min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// tiles of stream w of a partition with n_tiles tiles
__host__ __device__ __forceinline__ uint32_t sk_stream_tiles(uint32_t n_tiles, uint32_t w) {
  return n_tiles / SK_STREAMS + (w < n_tiles % SK_STREAMS ? 1u : 0u);
}
// first 1-KiB chunk of stream w inside the partition block (cpt = M/16 chunks per tile)
__host__ __device__ __forceinline__ uint32_t sk_stream_chunk0(uint32_t n_tiles, uint32_t w, uint32_t cpt) {
  const uint32_t q = n_tiles / SK_STREAMS, r = n_tiles % SK_STREAMS;
  const uint32_t tiles_before = q * w + sk_min_u32(w, r);
  const uint32_t nonempty_before = q ? w : sk_min_u32(w, r);
  return cpt * tiles_before + SK_TAIL_CHUNKS * nonempty_before;
}
// 1-KiB chunks of a whole partition block
__host__ __device__ __forceinline__ uint64_t sk_part_chunks(uint32_t n_tiles, uint32_t cpt) {
  return (uint64_t)cpt * n_tiles + SK_TAIL_CHUNKS * sk_min_u32(SK_STREAMS, n_tiles);
}
__host__ __device__ __forceinline__ bool sk_supported_m(uint32_t m) {
  return m == 32 || m == 48 || m == 64 || m == 80 || m == 96;
}
__host__ __device__ __forceinline__ uint32_t sk_pitch_dwords(uint32_t m) { return ((m + 32u + 31u) / 32u) * 32u; }

// ------------------------------------------------------------ index packing --
// Destination-driven: one 256-thread block per (partition, slot g).  Slots
// g < n_tiles write the M/16 chunks of tile g (stream g%16, position g/16): the
// bytes come from tile g (lanes already on it) and from the stream's previous
// tile g-16 (lanes still finishing it).  Slots g >= n_tiles write the 2 tail
// chunks of stream g - n_tiles.  Source tiles are staged through LDS so that
// both the reads (either source layout) and the 16-B writes are coalesced.
struct SkewPackArgs {
  const uint8_t* src;         // base of the staged chunk
  const uint64_t* src_off;    // [n_parts] byte offset of each partition inside src
  const uint32_t* part_ids;   // [n_parts]
  uint8_t* dst;
  const uint64_t* code_off;   // [nlist] byte offset of the partition block in dst
  const uint32_t* plen;
  uint32_t m;
  uint32_t transposed;        // source layout: 1 = [m][len] per partition, 0 = [len][m]
};

__global__ __launch_bounds__(256) void k_pack_skew(SkewPackArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tile[];  // [2][64][m+1]
  const uint32_t p = a.part_ids[blockIdx.y];
  const uint32_t len = a.plen[p];
  const uint32_t n_tiles = (len + SK_TILE - 1) / SK_TILE;
  const uint32_t n_slots = n_tiles + sk_min_u32(SK_STREAMS, n_tiles);
  const uint32_t g = blockIdx.x;
  if (g >= n_slots) return;
  const uint32_t m = a.m, pitch = m + 1, cpt = m / 16;
  const bool tail = g >= n_tiles;
  const uint32_t w = tail ? g - n_tiles : g % SK_STREAMS;
  const uint32_t nt_w = sk_stream_tiles(n_tiles, w);
  const uint32_t n = tail ? nt_w : g / SK_STREAMS;  // tile position inside the stream
  const uint8_t* src = a.src + a.src_off[blockIdx.y];
  // stage tile n (slot 0) and tile n-1 (slot 1) of stream w
  for (uint32_t which = 0; which < 2; ++which) {
    if (which == 0 && tail) continue;
    if (which == 1 && n == 0) continue;
    const uint32_t tg = w + SK_STREAMS * (n - which);  // global tile index
    const uint32_t r0 = tg * SK_TILE;
    unsigned char* t = tile + (size_t)which * 64u * pitch;
    for (uint32_t e = threadIdx.x; e < 64u * m; e += 256) {
      uint32_t i, j;
      if (a.transposed) {
        j = e / 64u;
        i = e % 64u;
      } else {
        i = e / m;
        j = e % m;
      }
      uint32_t v = 0;
      if (r0 + i < len) v = a.transposed ? src[(size_t)j * len + r0 + i] : src[(size_t)(r0 + i) * m + j];
      t[i * pitch + j] = (unsigned char)v;
    }
  }
  __syncthreads();
  const uint32_t n_chunks = tail ? SK_TAIL_CHUNKS : cpt;
  uint8_t* dst = a.dst + a.code_off[p] + ((size_t)sk_stream_chunk0(n_tiles, w, cpt) + (size_t)cpt * n) * 1024u;
  for (uint32_t e = threadIdx.x; e < n_chunks * 64u; e += 256) {
    const uint32_t cc = e / 64u, l = e % 64u, lm = l & 31u;
    uint32_t wds[4];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      uint32_t word = 0;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const uint32_t t = cc * 16u + k4 * 4 + kb;  // step inside this tile position
        uint32_t v = 0;
        if (t >= lm) {  // lane already on tile n: column j = t - lm
          if (!tail) v = tile[l * pitch + (t - lm)];
        } else if (n > 0) {  // still on tile n-1: column j = m + t - lm
          v = tile[(size_t)64u * pitch + l * pitch + (m + t - lm)];
        }
        word |= v << (8 * kb);
      }
      wds[k4] = word;
    }
    *(uint4*)(dst + (size_t)cc * 1024u + l * 16u) = make_uint4(wds[0], wds[1], wds[2], wds[3]);
  }
}

// codebook [m][256][dsub] -> [256][m][dsub]: the LUT builder then walks
// consecutive j with consecutive lanes (coalesced reads, conflict-free writes)
__global__ void k_transpose_codebook(const float* __restrict__ cb, uint32_t m, uint32_t dsub,
                                     float* __restrict__ out) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;  // over m*256*dsub
  const uint32_t total = m * 256u * dsub;
  if (e >= total) return;
  const uint32_t t = e % dsub, c = (e / dsub) % 256u, j = e / (dsub * 256u);
  out[((size_t)c * m + j) * dsub + t] = cb[e];
}

// ------------------------------------------------------------ work planning --
// Per batch: count the probing queries of every partition, lay the (query,
// probe rank) pairs out partition-major in the index's static partition order
// (grouped by XCD queue, longest partitions first) and reset the queue heads.
struct PlanArgs {
  const uint32_t* probes;   // [n_pairs]
  uint32_t n_pairs;
  uint32_t nlist;
  const uint32_t* plen;     // [nlist]
  const uint32_t* order;    // [nlist] static partition order
  const uint32_t* xcd_first;  // [9] index into order where queue x starts
  uint32_t* cnt;            // [nlist] (zeroed by k_plan_scan for the next batch)
  uint32_t* off;            // [nlist]
  uint32_t* fill;           // [nlist]
  uint32_t* q_start;        // [9]
  uint32_t* heads;          // [8 * SK_HEAD_STRIDE]
  uint32_t* items;          // [n_pairs]
  Cand* cand;               // [n_pairs][kk]
  uint32_t kk;
};

__global__ void k_plan_count(PlanArgs a) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_pairs) return;
  const uint32_t p = a.probes[i];
  if (a.plen[p])
    atomicAdd(&a.cnt[p], 1u);
  else {  // empty / not owned here: no work item, the slot is empty
    Cand c;
    c.d = __builtin_huge_valf();
    c.pos = CAND_EMPTY_POS;
    c.id = ~0ull;
    for (uint32_t g = 0; g < a.kk; ++g) a.cand[(size_t)i * a.kk + g] = c;
  }
}

// one 1024-thread block: exclusive scan of cnt[] in `order`
__global__ __launch_bounds__(1024) void k_plan_scan(PlanArgs a) {
  __shared__ uint32_t s_part[1024];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (a.nlist + 1023u) / 1024u;
  const uint32_t i0 = tid * per, i1 = sk_min_u32(a.nlist, i0 + per);
  uint32_t sum = 0;
  for (uint32_t i = i0; i < i1; ++i) sum += a.cnt[a.order[i]];
  s_part[tid] = sum;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
    uint32_t v = tid >= d ? s_part[tid - d] : 0u;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  uint32_t run = s_part[tid] - sum;
  for (uint32_t i = i0; i < i1; ++i) {
    const uint32_t p = a.order[i];
    const uint32_t c = a.cnt[p];
    a.off[p] = run;
    a.fill[p] = 0;
    a.cnt[p] = 0;
    // queue boundaries: the first partition of each queue records its offset
    for (uint32_t x = 0; x < 8; ++x)
      if (a.xcd_first[x] == i) a.q_start[x] = run;
    run += c;
  }
  if (tid == 1023) {
    const uint32_t total = s_part[1023];
    a.q_start[8] = total;
    for (uint32_t x = 0; x < 8; ++x)
      if (a.xcd_first[x] >= a.nlist) a.q_start[x] = total;
  }
  if (tid < 8) a.heads[tid * SK_HEAD_STRIDE] = 0;
}

__global__ void k_plan_fill(PlanArgs a) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_pairs) return;
  const uint32_t p = a.probes[i];
  if (!a.plen[p]) return;
  a.items[a.off[p] + atomicAdd(&a.fill[p], 1u)] = i;
}

// ------------------------------------------------------------------- scan ----
struct SkewArgs {
  IndexView ix;
  const float* cbT;         // [256][m][dsub]
  const float* qp;          // [nq, dim] preprocessed queries
  const uint32_t* probes;   // [nq * nprobe]
  const uint32_t* items;    // [n_items] pair index b*nprobe + r, partition-major
  const uint32_t* q_start;  // [9]
  uint32_t* heads;          // [8 * SK_HEAD_STRIDE]
  uint32_t* qthr;           // [nq] running per-query threshold (f32 sort key)
  uint32_t nprobe, kk;
  RangeFilter range;
  Cand* cand;               // [nq * nprobe][kk]
  uint32_t dbg;
};

__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}

#include "skew_chunks.inc"

// plain chunks g = G .. CPT-1 of a tile (steps >= 32)
template <int G, int CPT>
__device__ __forceinline__ void skew_plain_chunks(const uint4 (&cv)[CPT], uint32_t lb, uint32_t pb, float& x,
                                                  float& y) {
  if constexpr (G < CPT) {
    skew_chunk_plain<64 * G>(cv[G], lb, pb, x, y);
    skew_plain_chunks<G + 1, CPT>(cv, lb, pb, x, y);
  }
}

template <int M, int LR, int NT>
__global__ __launch_bounds__(NT) void k_scan_skew(SkewArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NW = NT / MI355_WAVE;
  constexpr int P = ((M + 32 + 31) / 32) * 32;  // LUT pitch in dwords
  constexpr int PB = P * 4;
  constexpr int CPT = M / 16;                   // 1-KiB chunks per tile
  const IndexView& ix = a.ix;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t lm = lane & 31;
  float* lut = (float*)smem;                                  // [256][P]
  float* res = (float*)(smem + 256 * PB);                     // [dim]
  ListEnt* lists = (ListEnt*)(smem + 256 * PB + (((size_t)ix.dim * 4 + 15) & ~(size_t)15));
  uint32_t* s_cnt = (uint32_t*)(lists + (size_t)NW * LR * MI355_WAVE);  // [NW]
  uint32_t* s_item = s_cnt + NW;                              // [1]
  uint32_t* s_thr = s_item + 1;                               // [1] block threshold (sort key)
  const uint32_t lb = (uint32_t)(size_t)smem + 4u * (32u - lm);  // LDS address of this lane's column origin
  const uint32_t pb = (uint32_t)__builtin_amdgcn_readfirstlane(PB);
  const uint32_t xcd = xcc_id();
  const uint64_t* rid = ix.row_ids;
  const bool ranged = a.range.has_lower || a.range.has_upper;

  for (;;) {
    // ---- next work item: own XCD's queue first, then steal -----------------
    if (tid == 0) {
      uint32_t it = SK_NONE;
      for (uint32_t k = 0; k < 8 && it == SK_NONE; ++k) {
        const uint32_t x = (xcd + k) & 7u;
        const uint32_t q0 = a.q_start[x], n = a.q_start[x + 1] - q0;
        uint32_t* head = a.heads + x * SK_HEAD_STRIDE;
        if (__hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n) continue;
        const uint32_t i = atomicAdd(head, 1u);
        if (i < n) it = a.items[q0 + i];
      }
      *s_item = it;
    }
    __syncthreads();
    const uint32_t pair = *s_item;
    if (pair == SK_NONE) break;
    const uint32_t b = pair / a.nprobe;
    const uint32_t p = a.probes[pair];
    const uint32_t len = ix.plen[p];
    const uint32_t n_tiles = (len + SK_TILE - 1) / SK_TILE;
    Cand* out = a.cand + (size_t)pair * a.kk;
    const float* q = a.qp + (size_t)b * ix.dim;

    // ---- K2: residual + distance table, [code][column] with duplicated tail -
    if (ix.metric == MI355_METRIC_DOT) {
      for (uint32_t d = tid; d < ix.dim; d += NT) res[d] = q[d];
    } else {
      const float* c = ix.centroids + (size_t)p * ix.dim;
      for (uint32_t d = tid; d < ix.dim; d += NT) res[d] = q[d] - c[d];
    }
    if (tid == 0) *s_thr = __hip_atomic_load(a.qthr + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (!(a.dbg & 1u)) {
      const uint32_t dsub = ix.dsub;
      for (uint32_t e = tid; e < 256u * M; e += NT) {
        const uint32_t c = e / (uint32_t)M, j = e % (uint32_t)M;
        const float* cb = a.cbT + (size_t)e * dsub;
        const float* rj = res + j * dsub;
        float acc = 0.f;
        if ((dsub & 3u) == 0) {  // 16-B loads; same d-ascending chain
          for (uint32_t t = 0; t < dsub; t += 4) {
            const float4 cv4 = *(const float4*)(cb + t);
            const float4 rv4 = *(const float4*)(rj + t);
            if (ix.metric == MI355_METRIC_DOT) {
              acc = __fmaf_rn(rv4.x, cv4.x, acc);
              acc = __fmaf_rn(rv4.y, cv4.y, acc);
              acc = __fmaf_rn(rv4.z, cv4.z, acc);
              acc = __fmaf_rn(rv4.w, cv4.w, acc);
            } else {
              float d0 = rv4.x - cv4.x, d1 = rv4.y - cv4.y, d2 = rv4.z - cv4.z, d3 = rv4.w - cv4.w;
              acc = __fmaf_rn(d0, d0, acc);
              acc = __fmaf_rn(d1, d1, acc);
              acc = __fmaf_rn(d2, d2, acc);
              acc = __fmaf_rn(d3, d3, acc);
            }
          }
        } else if (ix.metric == MI355_METRIC_DOT) {
          for (uint32_t t = 0; t < dsub; ++t) acc = __fmaf_rn(rj[t], cb[t], acc);
        } else {
          for (uint32_t t = 0; t < dsub; ++t) {
            float df = rj[t] - cb[t];
            acc = __fmaf_rn(df, df, acc);
          }
        }
        if (ix.metric == MI355_METRIC_DOT) acc = 1.0f - acc;
        lut[c * P + j + 32] = acc;
        if (j >= (uint32_t)(M - 31)) lut[c * P + j - (M - 32)] = acc;
      }
    }
    __syncthreads();

    // ---- K3 + K4: skewed ADC scan, one stream per wave ----------------------
    WaveList<LR> wl;
    wl.init(lists + (size_t)wid * LR * MI355_WAVE, a.kk);
    const uint32_t lrow0 = ix.lrow0[p];
    const uint64_t grow0 = ix.grow0[p];
    auto idof = [&](uint32_t pos) -> uint64_t { return rid ? rid[pos] : grow0 + (pos - lrow0); };
    float thr = f32_from_sort_key(*s_thr);
    if (*s_thr == 0xFFFFFFFFu) thr = __builtin_huge_valf();
    const uint8_t* pcodes = ix.codes + ix.code_off[p];

    // a finished row: tile position tp of stream w, this lane's row
    float published = __builtin_huge_valf();
    auto consume = [&](float acc, uint32_t w, uint32_t tp) {
      const uint32_t row = (w + SK_STREAMS * tp) * SK_TILE + lane;
      const float d = finalize_dist(acc, ix.metric, ix.m);
      bool ok = row < len && (ranged ? in_range(d, a.range) : d == d);
      // tightened by the other waves' compactions
      const uint32_t bk = __hip_atomic_load(s_thr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (bk != 0xFFFFFFFFu) thr = fminf(thr, f32_from_sort_key(bk));
      ok = ok && d <= thr;
      if (__any(ok)) {
        wl.append(ok, d, lrow0 + row, thr, lane, idof);
        if (wl.t_run < published) {  // a compaction tightened this wave's kk-th best: share it
          published = wl.t_run;
          if (lane == 0) atomicMin(s_thr, f32_sort_key(published));
        }
      }
    };

    for (uint32_t w = wid; w < SK_STREAMS && !(a.dbg & 2u); w += NW) {
      const uint32_t nt = sk_stream_tiles(n_tiles, w);
      if (!nt) continue;
      const uint4* src = (const uint4*)(pcodes + (size_t)sk_stream_chunk0(n_tiles, w, CPT) * 1024u) + lane;
      // two register sets, ping-pong: tile n is scanned from one while tile n+1
      // (or the 2 tail chunks) streams into the other
      uint4 ca[CPT], cb[CPT];
      auto fetch = [&](uint4 (&dst)[CPT], uint32_t n) {  // n == nt: the tail chunks
        const uint4* s2 = src + (size_t)n * CPT * 64;
        if (n < nt) {
#pragma unroll
          for (int g = 0; g < CPT; ++g) dst[g] = s2[(size_t)g * 64];
        } else {
#pragma unroll
          for (int g = 0; g < (int)SK_TAIL_CHUNKS; ++g) dst[g] = s2[(size_t)g * 64];
        }
      };
      float x = 0.f, y = 0.f;
      auto tile = [&](const uint4 (&cv)[CPT], uint32_t n) {
        skew_chunk_split0<0>(cv[0], lb, pb, x, y);
        skew_chunk_split1<64>(cv[1], lb, pb, x, y);
        if (n > 0) consume(y, w, n - 1);  // row n-1 is complete on every lane after step 30
        skew_plain_chunks<2, CPT>(cv, lb, pb, x, y);
        y = x;
        x = 0.f;
      };
      auto tails = [&](const uint4 (&cv)[CPT]) {  // 31 more steps finish the last tile's rows
        float dummy = 0.f;
        skew_chunk_split0<0>(cv[0], lb, pb, dummy, y);
        skew_chunk_split1<64>(cv[1], lb, pb, dummy, y);
        consume(y, w, nt - 1);
      };
      fetch(ca, 0);
      for (uint32_t n = 0;;) {
        fetch(cb, n + 1);
        tile(ca, n);
        if (++n == nt) {
          tails(cb);
          break;
        }
        fetch(ca, n + 1);
        tile(cb, n);
        if (++n == nt) {
          tails(ca);
          break;
        }
      }
    }

    // ---- block result: exact kk best of all waves' lists, written sorted ----
    if (wl.cnt > a.kk) wl.compact(lane, idof);
    if (lane == 0) s_cnt[wid] = wl.cnt;
    __syncthreads();
    uint32_t total = 0;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) total += s_cnt[w2];
    const uint32_t n_out = min(total, a.kk);
    for (uint32_t g = tid; g < (uint32_t)NW * a.kk; g += NT) {
      const uint32_t w = g / a.kk, j = g % a.kk;
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
      if (rank < a.kk) {
        Cand o;
        o.d = mine.d;
        o.pos = mine.pos;
        o.id = idof(mine.pos);
        out[rank] = o;
        // kk rows at or below mine.d exist: a bound for every other partition of this query
        if (rank == a.kk - 1) atomicMin(a.qthr + b, f32_sort_key(mine.d));
      }
    }
    for (uint32_t g = n_out + tid; g < a.kk; g += NT) {
      Cand o;
      o.d = __builtin_huge_valf();
      o.pos = CAND_EMPTY_POS;
      o.id = ~0ull;
      out[g] = o;
    }
    __syncthreads();  // LDS is rebuilt by the next item
  }
}
