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
#include <type_traits>

#include "kernels_ivfpq.h"
#include "skew_chunks.inc"  // generated inner blocks; defines SK_ADDR_* / SK_SPLIT_*
typedef __attribute__((ext_vector_type(4))) float sk_f32x4;

#ifndef SK_LUT_INFLIGHT
#define SK_LUT_INFLIGHT 8  // 16-B codebook loads in flight per thread while the distance table is built
#endif
#ifndef SK_RING_FULL
#define SK_RING_FULL 0  // dev knob: 1 = ring of a whole tile's chunks (prefetch distance CPT-1 chunks)
#endif
#ifdef SK_DUAL
#define SK_CHAINS 2u  // rows in flight per lane (streams per wave)
#else
#define SK_CHAINS 1u
#endif
#define SK_UNITS 16u                        // units per partition = waves of the 1024-thread scan workgroup
#define SK_STREAMS (SK_UNITS * SK_CHAINS)   // tile g belongs to stream g % SK_STREAMS
#define SK_TILE 64u
#define SK_TAIL_CHUNKS 2u  // 32 skew steps / 16 steps per chunk
#define SK_NONE 0xFFFFFFFFu
#define SK_HEAD_STRIDE 32u  // u32 words between the per-XCD queue heads (128 B)

// One scan work item, self-contained so that a workgroup needs ONE dependent
// load after the queue pop (written per batch by k_plan_fill).
struct __attribute__((aligned(32))) SkewItem {
  uint32_t pair;      // b * nprobe + r (indexes probes[] and the candidate slots)
  uint32_t part;      // partition id
  uint32_t len;       // rows of the partition on this handle
  uint32_t lrow0;     // local position of its first row
  uint64_t grow0;     // global index position of its first row
  uint64_t code_off;  // byte offset of its code block
};

__host__ __device__ __forceinline__ uint32_t sk_min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// SkewItem::pair of a SLICED batch (SkewArgs::n_slices > 1): [31:26] slices of this pair - 1, [25:20] this item's
// slice, [19:0] the pair.  (The slice count travels with the item: a planner may cut pairs unevenly — round 5 measured
// cutting by rows, no gain, NOTES 10.3 — without touching the kernel.)
#define SK_MAX_SLICES 64u
__host__ __device__ __forceinline__ uint32_t sk_pack_pair(uint32_t pair, uint32_t slice, uint32_t n_sl) {
  return pair | (slice << 20) | ((n_sl - 1u) << 26);
}

// phase (steps of delay) of lane l: distinct inside each 32-lane bank group
__host__ __device__ __forceinline__ uint32_t sk_phase(uint32_t l) {
#ifdef SK_SPLIT_BFM
  return l < 32u ? l : 63u - l;  // mirrored: the lanes still on the old row are contiguous (s_bfm_b64)
#else
  return l & 31u;
#endif
}

__device__ __forceinline__ void sk_rotate_prio(uint32_t turn) {  // (s_setprio takes an immediate)
  switch (turn & 3u) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
}

// Unit w of a partition = the SK_CHAINS streams w*SK_CHAINS .. scanned together by
// one wave.  Its chains are padded to the tile count of its first stream.
__host__ __device__ __forceinline__ uint32_t sk_unit_tiles(uint32_t n_tiles, uint32_t w) {
  return n_tiles / SK_STREAMS + (w * SK_CHAINS < n_tiles % SK_STREAMS ? 1u : 0u);
}
// first 1-KiB chunk of unit w inside the partition block (cpt = M/16 chunks per tile);
// a unit stores (cpt * tiles + 2 tail) groups of SK_CHAINS chunks (one per chain)
__host__ __device__ __forceinline__ uint32_t sk_unit_chunk0(uint32_t n_tiles, uint32_t w, uint32_t cpt) {
  const uint32_t q = n_tiles / SK_STREAMS, r = n_tiles % SK_STREAMS;
  const uint32_t longer = (r + SK_CHAINS - 1) / SK_CHAINS;  // units holding one tile more
  const uint32_t tiles_before = q * w + sk_min_u32(w, longer);
  const uint32_t nonempty_before = q ? w : sk_min_u32(w, longer);
  return SK_CHAINS * (cpt * tiles_before + SK_TAIL_CHUNKS * nonempty_before);
}
// 1-KiB chunks of a whole partition block
__host__ __device__ __forceinline__ uint64_t sk_part_chunks(uint32_t n_tiles, uint32_t cpt) {
  const uint32_t q = n_tiles / SK_STREAMS, r = n_tiles % SK_STREAMS;
  const uint32_t longer = (r + SK_CHAINS - 1) / SK_CHAINS;
  const uint64_t unit_tiles = (uint64_t)q * SK_UNITS + longer;
  const uint32_t nonempty = q ? SK_UNITS : longer;
  return (uint64_t)SK_CHAINS * (cpt * unit_tiles + SK_TAIL_CHUNKS * nonempty);
}
// grid.x of k_pack_skew for partitions of up to max_rows rows
__host__ __device__ __forceinline__ uint32_t sk_pack_slots(uint32_t max_rows) {
  const uint32_t n_tiles = (max_rows + SK_TILE - 1) / SK_TILE;
  return ((n_tiles + SK_STREAMS - 1) / SK_STREAMS + 1u) * SK_STREAMS;
}
// LDS bytes of the distance table: 256 codes x 128 columns (dual: slab 1 starts one row late)
#ifdef SK_DUAL
#define SK_TABLE_BYTES (131072u + 256u)
#else
#define SK_TABLE_BYTES 131072u
#endif
// Table widths the kernel is instantiated for (LUT columns per slab): the table holds M + 32 <= 128 columns.
__host__ __device__ __forceinline__ bool sk_kernel_m(uint32_t m) {
  return m == 32 || m == 48 || m == 64 || m == 80 || m == 96;
}
// Every 8-bit m the reference's builder produces (index/vector.rs:306-319: dim / 16, dim / 8 or 1) runs this kernel:
//  * m <= 96 that is not a kernel width is PADDED to the next one with code 0 and an all-zero table column — the
//    row sum stays the contract's j-ascending chain followed by `+ 0.0f` terms, which are exact;
//  * m > 96 is cut into n_slabs slabs of M columns (the last one padded): a work item builds slab s's table, scans
//    slab s's code streams starting every row's accumulator from the row's partial sum of slabs 0..s-1 (parked in a
//    per-workgroup scratch, 4 B per row against M code bytes) and selects in the last slab — still j-ascending.
struct SkewShape {
  uint32_t M;        // columns per slab = template width of k_scan_skew
  uint32_t n_slabs;  // slabs per row
  uint32_t slabbed;  // 1: the generalised kernel (padding and / or several slabs); 0: m == M, one slab
};
#define SK_MAX_SLABS 8u
__host__ __device__ __forceinline__ bool sk_shape(uint32_t m, SkewShape* out) {
  if (m == 0 || m > 96u * SK_MAX_SLABS) return false;
  const uint32_t n_slabs = (m + 95u) / 96u;
  const uint32_t per = (m + n_slabs - 1) / n_slabs;
  uint32_t M = (per + 15u) & ~15u;
  if (M < 32u) M = 32u;  // a tile is at least the 32 skew steps
  out->M = M;
  out->n_slabs = n_slabs;
  out->slabbed = (n_slabs > 1u || M != m) ? 1u : 0u;
  return true;
}
// dynamic LDS of a scan workgroup: table + residual (res_floats f32) + nw lists of lr * 64 entries + small words
// LDS bytes of a table of M columns per slab: table columns u = 1 .. M + 31; up to M = 32 they all sit in slab 0
__host__ __device__ constexpr uint32_t sk_table_bytes(uint32_t M) { return M <= 32u ? 65536u : SK_TABLE_BYTES; }
__host__ __device__ __forceinline__ size_t sk_scan_lds(uint32_t M, uint32_t res_floats, int nw, int lr) {
  return (size_t)sk_table_bytes(M) + (((size_t)res_floats * 4 + 15) & ~(size_t)15) + (size_t)nw * lr * 64 * 8 +
         (size_t)(2 * nw + 11) * 4 + 128;
}
// LUT pitch in dwords: >= m + 32 columns and a power of two, so that the address
// is (code << 9) | column bytes (three 2-cycle VOP2 ops, scripts/gen_skew_chunks.py)
__host__ __device__ __forceinline__ uint32_t sk_pitch_dwords(uint32_t) { return 128u; }

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
  uint32_t m;                 // columns per slab (SkewShape::M): the packed row width
  uint32_t m_src;             // columns per source row (the index's m); slab z packs columns z*m .. of it, zero-padded
  uint32_t transposed;        // source layout: 1 = [bytes per row][len] per partition, 0 = [len][bytes per row]
  // 4: 4-bit PQ (table/create_index.rs:86-102) — a source row is m_src / 2 bytes, sub-quantiser 2t in the low nibble of byte t; the
  // packed streams hold one BYTE per column (values 0..15): the scan is the 8-bit one against a 16-row table, at twice the stored
  // bytes of the packed-nibble form and 2.5 x the speed of the generic 4-bit kernel (round 4: 30 k -> see DESIGN.md section 4.1)
  uint32_t nbits;
};

// grid = (slots, partitions of the batch, slabs); a partition block holds its slabs one after the other, each laid
// out as a partition of m-byte rows
static __global__ __launch_bounds__(256) void k_pack_skew(SkewPackArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tile[];  // [2][64][m+1]
  const uint32_t p = a.part_ids[blockIdx.y];
  const uint32_t len = a.plen[p];
  const uint32_t n_tiles = (len + SK_TILE - 1) / SK_TILE;
  // slot -> (position n inside the unit, unit w, chain ch); n == unit tiles is the tail
  const uint32_t slot = blockIdx.x;
  const uint32_t st = slot % SK_STREAMS, n = slot / SK_STREAMS;
  const uint32_t w = st / SK_CHAINS, ch = st % SK_CHAINS;
  const uint32_t nt_w = sk_unit_tiles(n_tiles, w);
  if (nt_w == 0 || n > nt_w) return;
  const uint32_t m = a.m, pitch = m + 1, cpt = m / 16;
  const uint32_t j0 = blockIdx.z * m;  // first source column of this slab
  const bool tail = n == nt_w;
  const uint8_t* src = a.src + a.src_off[blockIdx.y];
  // stage tile position n (slot 0) and n-1 (slot 1) of this chain's stream; positions past
  // the stream's end (chains padded to the unit's tile count) are all padding rows
  for (uint32_t which = 0; which < 2; ++which) {
    if (which == 0 && tail) continue;
    if (which == 1 && n == 0) continue;
    const uint32_t tg = st + SK_STREAMS * (n - which);  // global tile index
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
      if (r0 + i < len && j0 + j < a.m_src) {
        const uint32_t col = j0 + j;
        if (a.nbits == 4) {
          const uint32_t byte = col >> 1, mbs = a.m_src >> 1;
          const uint32_t bv = a.transposed ? src[(size_t)byte * len + r0 + i] : src[(size_t)(r0 + i) * mbs + byte];
          v = (bv >> (4u * (col & 1u))) & 15u;
        } else {
          v = a.transposed ? src[(size_t)col * len + r0 + i] : src[(size_t)(r0 + i) * a.m_src + col];
        }
      }
      t[i * pitch + j] = (unsigned char)v;
    }
  }
  __syncthreads();
  const uint32_t n_chunks = tail ? SK_TAIL_CHUNKS : cpt;
  uint8_t* dst = a.dst + a.code_off[p] + (size_t)blockIdx.z * sk_part_chunks(n_tiles, cpt) * 1024u +
                 (size_t)sk_unit_chunk0(n_tiles, w, cpt) * 1024u;
  for (uint32_t e = threadIdx.x; e < n_chunks * 64u; e += 256) {
    const uint32_t cc = e / 64u, l = e % 64u, lm = sk_phase(l);
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
    // chunk (n, cc) of the unit: SK_CHAINS consecutive 1-KiB chunks, one per chain
    const size_t chunk = ((size_t)n * cpt + cc) * SK_CHAINS + ch;
    *(uint4*)(dst + chunk * 1024u + l * 16u) = make_uint4(wds[0], wds[1], wds[2], wds[3]);
  }
}

// codebook [m][256][dsub] -> [256][m][dsub]: the LUT builder then walks
// consecutive j with consecutive lanes (coalesced reads, conflict-free writes)
static __global__ void k_transpose_codebook(const float* __restrict__ cb, uint32_t m, uint32_t dsub, uint32_t ksub,
                                     float* __restrict__ out) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;  // over m*ksub*dsub (ksub = 256, or 16 for 4-bit PQ)
  const uint32_t total = m * ksub * dsub;
  if (e >= total) return;
  const uint32_t t = e % dsub, c = (e / dsub) % ksub, j = e / (dsub * ksub);
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
  const uint32_t* opos;     // [nlist] its inverse: position of partition p in `order`
  const uint32_t* xcd_first;  // [9] index into order where queue x starts
  uint32_t* cnt;            // [2 * nlist] indexed by plan_vkey (zeroed by k_plan_scan for the next batch)
  uint32_t* off;            // [2 * nlist] (same index)
  uint32_t* fill;           // [2 * nlist] (same index)
  uint32_t* q_start;        // [9]
  uint32_t* heads;          // [8 * SK_HEAD_STRIDE]
  SkewItem* items;          // [n_pairs]
  const uint32_t* lrow0;    // [nlist]
  const uint64_t* grow0;    // [nlist]
  const uint64_t* code_off; // [nlist]
  uint32_t* cand_cnt;       // [n_pairs] rows each work item left in its kk candidate slots (0 until it ran)
  uint32_t kk;
  uint32_t nprobe;          // pairs per query (for the mask)
  // 1: every queue runs the (query, NEAREST partition) items first — probe rank 0 of every query,
  // k_select_probes puts the nearest partition there — and the partition-major rest behind them.  The
  // nearest partition's kk-th best is the tightest bound a single partition can give (qthr), so the
  // other 63 items of a query admit a handful of rows instead of filling their lists: what long
  // candidate lists cost is selection, and selection work follows the rows admitted.  The price is one
  // extra, un-shared read of ~one partition per query (+13 % L2 fills at C3, HBM is 15 % busy).
  uint32_t best_first;
  uint32_t n_slices;        // work items per pair (SkewArgs::n_slices); with by_rows: the most a pair is cut into (its slot stride)
  // by_rows > 0 (the sparse planner only): pairs are cut by ROWS, not by count — a pair of P tile positions becomes ceil(P / T) work
  // items, T the smallest number of positions per item that leaves at most n_wg * by_rows items (round 6: with eight slices per
  // pair whatever its length, the eight workgroups on a single query's longest partition — 2.8 x the mean at C3's skew — ran 75 us
  // items and then took a second one, while the first workgroup was out of work after 57 of the launch's 92 us)
  uint32_t by_rows;         // target work items per scan workgroup (0: n_slices items per pair)
  uint32_t n_wg;            // scan workgroups of the launch
  ActiveMask act;           // device-side batch size: pairs of inactive queries make no item, no slot writes
};

// work-item class of a pair: 1 = the query's nearest partition under best_first
__device__ __forceinline__ uint32_t plan_class(const PlanArgs& a, uint32_t i) {
  return (a.best_first && (i % a.nprobe) == 0u) ? 1u : 0u;
}
// cnt / off / fill are indexed by a (partition, class)'s place in the VIRTUAL sequence the work list is laid out in: queue by
// queue, [class-1 counts of the queue's partitions (best_first only)] [class-0 counts of the same partitions], partitions in
// the index's static `order`.  (Rounds 4-6 indexed them by partition + class * nlist and k_plan_scan walked the sequence
// through `order`: two dependent, uncoalesced loads and three scattered stores per counter from ONE workgroup — bound by the
// CU's one-line-per-clock address path, 80 us at the reference's default 12 207 partitions.)
__device__ __forceinline__ uint32_t plan_vkey(const PlanArgs& a, uint32_t p, uint32_t cls) {
  const uint32_t pos = a.opos[p];
  if (!a.best_first) return pos;
  uint32_t x = 0;
  for (uint32_t y = 1; y < 8; ++y) x += (pos >= a.xcd_first[y]) ? 1u : 0u;  // queue of p (empty queues are skipped over)
  const uint32_t x0 = a.xcd_first[x], len = a.xcd_first[x + 1] - x0, idx = pos - x0;
  return 2u * x0 + (cls ? idx : len + idx);
}

__device__ __forceinline__ void plan_count_pair(const PlanArgs& a, uint32_t i) {
  if (!a.act.on(i / a.nprobe)) return;
  const uint32_t p = a.probes[i];
  // ids outside the index (mi355_search_probes), empty and not-owned partitions make no work item; a
  // pair's candidate slots hold cand_cnt[i] rows: 0 until (and unless) its item ran
  for (uint32_t sl = 0; sl < a.n_slices; ++sl) a.cand_cnt[(size_t)i * a.n_slices + sl] = 0u;
  if (p < a.nlist && a.plen[p]) atomicAdd(&a.cnt[plan_vkey(a, p, plan_class(a, i))], a.n_slices);
}
static __global__ void k_plan_count(PlanArgs a) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.n_pairs) plan_count_pair(a, i);
}

// one 1024-thread block: exclusive scan of the item counts in the virtual sequence (= index order of cnt / off / fill), in
// chunks of PLAN_CHUNK counters staged through LDS: coalesced loads, every thread scans its eight consecutive counters, the
// chunk's offsets leave coalesced again (a chunk is one memory round trip + one block scan: ~3 us).
#define PLAN_CHUNK 8192u
#define PLAN_CHUNK_WORDS (PLAN_CHUNK + PLAN_CHUNK / 8u)  // (a thread's eight counters start 9 words apart: no bank conflicts)
__device__ __forceinline__ void plan_scan_block(const PlanArgs& a, uint32_t* s_part /*[1024]*/, uint32_t* s_xf /*[9]*/,
                                                uint32_t* s_c /*[PLAN_CHUNK_WORDS]*/) {
  const uint32_t tid = threadIdx.x;
  const uint32_t ncls = a.best_first ? 2u : 1u;
  const uint32_t nv = ncls * a.nlist;
  if (tid < 9) s_xf[tid] = a.xcd_first[tid];
  __syncthreads();
  auto slot = [](uint32_t i) -> uint32_t { return i + (i >> 3); };
  uint32_t carry = 0;
  for (uint32_t b = 0; b < nv; b += PLAN_CHUNK) {
    uint32_t c[8];
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) {
      const uint32_t v = b + u * 1024u + tid;
      c[u] = v < nv ? a.cnt[v] : 0u;
    }
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) s_c[slot(u * 1024u + tid)] = c[u];
    __syncthreads();
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) {
      c[u] = s_c[slot(tid * 8u + u)];
      sum += c[u];
    }
    s_part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
      uint32_t v = tid >= d ? s_part[tid - d] : 0u;
      __syncthreads();
      s_part[tid] += v;
      __syncthreads();
    }
    uint32_t run = carry + s_part[tid] - sum;
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) {
      s_c[slot(tid * 8u + u)] = run;
      run += c[u];
    }
    const uint32_t total = s_part[1023];
    __syncthreads();
#pragma unroll
    for (uint32_t u = 0; u < 8; ++u) {
      const uint32_t v = b + u * 1024u + tid;
      if (v < nv) {
        a.off[v] = s_c[slot(u * 1024u + tid)];
        a.fill[v] = 0;
        a.cnt[v] = 0;
      }
    }
    // queue boundaries: the first virtual index of each queue records its offset
    if (tid < 8) {
      const uint32_t v = ncls * s_xf[tid];
      if (v >= b && v < b + PLAN_CHUNK && v < nv) a.q_start[tid] = s_c[slot(v - b)];
    }
    carry += total;
    __syncthreads();
  }
  if (tid < 8 && s_xf[tid] >= a.nlist) a.q_start[tid] = carry;
  if (tid == 8) a.q_start[8] = carry;
  if (tid < 8) a.heads[tid * SK_HEAD_STRIDE] = 0;
}
static __global__ __launch_bounds__(1024) void k_plan_scan(PlanArgs a) {
  __shared__ uint32_t s_part[1024];
  __shared__ uint32_t s_xf[9];
  __shared__ uint32_t s_c[PLAN_CHUNK_WORDS];
  plan_scan_block(a, s_part, s_xf, s_c);
}

__device__ __forceinline__ void plan_fill_pair(const PlanArgs& a, uint32_t i) {
  if (!a.act.on(i / a.nprobe)) return;
  const uint32_t p = a.probes[i];
  const uint32_t len = p < a.nlist ? a.plen[p] : 0u;
  if (!len) return;
  SkewItem it;
  it.pair = i;
  it.part = p;
  it.len = len;
  it.lrow0 = a.lrow0[p];
  it.grow0 = a.grow0[p];
  it.code_off = a.code_off[p];
  const uint32_t key = plan_vkey(a, p, plan_class(a, i));
  const uint32_t at = a.off[key] + atomicAdd(&a.fill[key], a.n_slices);
  for (uint32_t sl = 0; sl < a.n_slices; ++sl) {
    it.pair = a.n_slices > 1u ? sk_pack_pair(i, sl, a.n_slices) : i;
    a.items[at + sl] = it;
  }
}
static __global__ void k_plan_fill(PlanArgs a) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.n_pairs) plan_fill_pair(a, i);
}
// The three steps as ONE workgroup for the batches of a few queries (a launch boundary costs more than the steps): the
// counts, offsets and fill cursors are only touched by this workgroup's own atomics / stores between its barriers
// (device-scope atomics execute at L2, stores write through, and the barrier waits for both), and no line of them
// was read earlier in the kernel.
#define PLAN_FUSED_MAX_PAIRS 4096u
static __global__ __launch_bounds__(1024) void k_plan_fused(PlanArgs a) {
  __shared__ uint32_t s_part[1024];
  __shared__ uint32_t s_xf[9];
  __shared__ uint32_t s_c[PLAN_CHUNK_WORDS];
  for (uint32_t i = threadIdx.x; i < a.n_pairs; i += 1024u) plan_count_pair(a, i);
  __threadfence();
  __syncthreads();
  plan_scan_block(a, s_part, s_xf, s_c);
  __threadfence();
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < a.n_pairs; i += 1024u) plan_fill_pair(a, i);
}

// A few queries (<= PLAN_SPARSE_MAX_PAIRS pairs): the same work list without walking the 2 * nlist counters — every
// pair computes its place in the virtual sequence directly (queue of its partition, class, position in `order`), the
// pairs are ranked by counting over (place, pair index), and the queue boundaries are counts of the places below them.
// Items of one partition and class land in pair-index order (the counting planner leaves that order to its atomics;
// placement affects speed only).  cnt / off / fill are not touched (cnt stays zeroed for the counting planner).
#define PLAN_SPARSE_MAX_PAIRS 512u
// (the body runs inside any workgroup of >= PLAN_SPARSE_MAX_PAIRS threads: k_plan_sparse alone, or the last workgroup of
//  k_select_plan.  FRESH: the probe lists were written by OTHER workgroups of the same launch — read them at L2.)
// tile positions of the longest unit of a partition of `len` rows: what a pair can be cut into (a slice is a range of positions)
__host__ __device__ __forceinline__ uint32_t sk_positions(uint32_t len) {
  const uint32_t n_tiles = (len + SK_TILE - 1) / SK_TILE;
  return (n_tiles + SK_STREAMS - 1) / SK_STREAMS;
}
#define PLAN_BY_ROWS_AUTO 0xFFFFFFFFu
#define PLAN_T_TRIES 4u  // candidate positions-per-item tried side by side (by_rows)
#define PLAN_SQ_BINS 13u   // s_q[13 ..]: 16 bins x {pairs, items, cursor}
#define PLAN_SQ_WORDS (PLAN_SQ_BINS + 48u)
template <bool FRESH>
__device__ __forceinline__ void plan_sparse_body(const PlanArgs& a, uint32_t* s_key /*[PLAN_SPARSE_MAX_PAIRS]*/, uint32_t* s_xf /*[9]*/,
                                                 uint32_t* s_q /*[PLAN_SQ_WORDS]*/,
                                                 const uint32_t* lds_probes = nullptr /*[n_pairs] in LDS: skip the global read*/,
                                                 unsigned long long* stat_rows = nullptr /*+= probed rows of the batch*/,
                                                 uint2* s_list = nullptr /*[PLAN_SPARSE_MAX_PAIRS] the pairs grouped by bin: {key, items << 16 | pair}*/) {
  const uint32_t i = threadIdx.x, lane = i & 63u;
  const uint32_t ncls = a.best_first ? 2u : 1u;
  const bool by_rows = a.by_rows != 0u && a.n_slices > 1u;
#ifdef MI355_DEV_PLAN
  const unsigned long long pl_t0 = wall_clock64();
  unsigned long long pl_t1 = 0, pl_t2 = 0, pl_t3 = 0;
#define PL_STAMP(x) x = wall_clock64()
#else
#define PL_STAMP(x)
#endif
  if (i < 9) s_xf[i] = a.xcd_first[i];
  // [9] probed rows, [10] tile positions, [11], [12] items at T0 + k, two per word, [13 ..] the bins ([0..8]: unused)
  if (i < PLAN_SQ_WORDS) s_q[i] = 0;
  __syncthreads();
  // (every thread loads unconditionally at clamped indices — its probe, then the five per-partition words side by side:
  //  guarded, each load is a memory round trip of its own and the item records waited for three more at the end)
  uint32_t key = 0xFFFFFFFFu, p = 0xFFFFFFFFu, len = 0, xq = 0;
  const bool mine = i < a.n_pairs && a.act.on(i / a.nprobe);
  const uint32_t ic = i < a.n_pairs ? i : a.n_pairs - 1u;
  const uint32_t p_ld = lds_probes ? lds_probes[ic]
                        : FRESH  ? __hip_atomic_load(a.probes + ic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                 : a.probes[ic];
  const uint32_t pc = p_ld < a.nlist ? p_ld : 0u;
  const uint32_t len_ld = a.plen[pc], at = a.opos[pc], it_lrow0 = a.lrow0[pc];
  const uint64_t it_grow0 = a.grow0[pc], it_code_off = a.code_off[pc];
  if (mine) {
    for (uint32_t sl = 0; sl < a.n_slices; ++sl) a.cand_cnt[(size_t)i * a.n_slices + sl] = 0u;
    p = p_ld;
    len = p < a.nlist ? len_ld : 0u;
    if (len) {
      for (uint32_t y = 1; y < 8; ++y) xq += (at >= s_xf[y]) ? 1u : 0u;  // queue of the partition (empty queues are skipped over)
      const uint32_t qlen = s_xf[xq + 1] - s_xf[xq], idx = at - s_xf[xq];
      key = ncls * s_xf[xq] + ((ncls == 2u && plan_class(a, i) == 0u) ? qlen : 0u) + idx;
    }
  }
  if (i < PLAN_SPARSE_MAX_PAIRS) s_key[i] = key;
  auto wave_sum = [&](uint32_t v) -> uint32_t {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v += (uint32_t)__shfl_xor((int)v, off);
    return v;
  };
  const bool live = key != 0xFFFFFFFFu;
  const uint32_t pos = live ? sk_positions(len) : 0u;
  if (stat_rows || by_rows) {  // probed rows / tile positions of the batch: wave sums, one LDS atomic per wave
    if (__any(live)) {
      const uint32_t v = wave_sum(live ? len : 0u), vp = by_rows ? wave_sum(pos) : 0u;
      if (lane == 0 && v) atomicAdd(&s_q[9], v);
      if (lane == 0 && vp) atomicAdd(&s_q[10], vp);
    }
    __syncthreads();
    if (stat_rows && i == 0 && s_q[9]) atomicAdd(stat_rows, (unsigned long long)s_q[9]);
  }
  PL_STAMP(pl_t1);
  // work items of this pair
  uint32_t nsl = live ? a.n_slices : 0u;
  if (by_rows) {
    // items per workgroup the launch should not exceed: an item costs its table (10 us) and its streams' tails (a third of a
    // position) before it scans anything, so a workgroup gets ONE item unless its share is long enough to hide that — two from
    // 8 positions, three from 12
    const uint32_t share = s_q[10] / max(a.n_wg, 1u);
    const uint32_t ipc = a.by_rows != PLAN_BY_ROWS_AUTO ? a.by_rows : min(3u, max(1u, share / 4u));
    const uint32_t cap = a.n_wg * ipc;
    const uint32_t t0 = max(1u, (s_q[10] + cap - 1u) / max(cap, 1u));
    // the items of the launch at T0 .. T0 + 3 positions per item, side by side (a pair gives <= 64 items, a wave <= 4096, the
    // batch <= 32768: 16-bit fields, two per word)
    if (__any(live)) {
      uint32_t c[PLAN_T_TRIES];
#pragma unroll
      for (uint32_t k = 0; k < PLAN_T_TRIES; ++k) c[k] = min((pos + t0 + k - 1u) / (t0 + k), a.n_slices);
      const uint32_t w01 = wave_sum(c[0] | (c[1] << 16)), w23 = wave_sum(c[2] | (c[3] << 16));
      if (lane == 0) {
        atomicAdd(&s_q[11], w01);
        atomicAdd(&s_q[12], w23);
      }
    }
    __syncthreads();
    const uint32_t n_at[PLAN_T_TRIES] = {s_q[11] & 0xFFFFu, s_q[11] >> 16, s_q[12] & 0xFFFFu, s_q[12] >> 16};
    uint32_t t = t0 + PLAN_T_TRIES - 1u;
#pragma unroll
    for (uint32_t k = PLAN_T_TRIES; k-- > 0;)
      if (n_at[k] <= cap) t = t0 + k;
    nsl = live ? max(1u, min((pos + t - 1u) / t, a.n_slices)) : 0u;
  }
  // The place of a pair's items: behind the items of the BINS below its own — (queue, class), class 1 (the nearest partitions) first, the
  // order of `key` — and behind the pairs of its own bin with a smaller key.  Counting over all pairs instead (every thread sweeping
  // 512 keys in LDS) was 13 of the 24 us a batch of 8 spent planning: the sweep is LDS-bandwidth-bound however it is split.
  uint32_t* s_bcnt = s_q + PLAN_SQ_BINS;         // [16] pairs per bin
  uint32_t* s_bitems = s_bcnt + 16;              // [16] items per bin
  uint32_t* s_bcur = s_bitems + 16;              // [16] scatter cursors
  const uint32_t bin = xq * 2u + ((ncls == 2u && plan_class(a, i) == 0u) ? 1u : 0u);
  if (live) {
    atomicAdd(&s_bcnt[bin], 1u);
    atomicAdd(&s_bitems[bin], nsl);
  }
  __syncthreads();
  uint32_t off_pairs = 0, off_items = 0;
  if (live) {
    for (uint32_t y = 0; y < bin; ++y) {
      off_pairs += s_bcnt[y];
      off_items += s_bitems[y];
    }
    s_list[off_pairs + atomicAdd(&s_bcur[bin], 1u)] = make_uint2(key, (nsl << 16) | i);  // (nsl <= 64, i < 512)
  }
  __syncthreads();
  PL_STAMP(pl_t2);
  uint32_t before = off_items;
  if (live) {
    // (the bin's members in a row, key and items side by side: a list of pair indices made every step two DEPENDENT LDS reads — 10 us
    //  for the 63 members of a batch of 8's bins)
    const uint32_t n_b = s_bcnt[bin];
    const uint2* mem = s_list + off_pairs;
#pragma unroll 4
    for (uint32_t t = 0; t < n_b; ++t) {
      const uint2 mj = mem[t];
      before += (mj.x < key || (mj.x == key && (mj.y & 0xFFFFu) < i)) ? (mj.y >> 16) : 0u;
    }
  }
  if (live) {
    PL_STAMP(pl_t3);
    SkewItem it;
    it.part = p;
    it.len = len;
    it.lrow0 = it_lrow0;
    it.grow0 = it_grow0;
    it.code_off = it_code_off;
    for (uint32_t sl = 0; sl < nsl; ++sl) {
      it.pair = a.n_slices > 1u ? sk_pack_pair(i, sl, nsl) : i;
      a.items[(size_t)before + sl] = it;
    }
  }
  if (i < 9) {  // queue x starts behind the items of the queues below it
    uint32_t acc = 0;
    for (uint32_t y = 0; y < 2u * i; ++y) acc += s_bitems[y];
    a.q_start[i] = acc;
  }
  if (i < 8) a.heads[i * SK_HEAD_STRIDE] = 0;
#ifdef MI355_DEV_PLAN  // dev[0..3] += loads + totals / T + histogram / rank loop / emit (thread 0's stamps; stat_rows = &DevCtl::rows_scanned)
  if (i == 0 && stat_rows) {
    uint32_t* dev = (uint32_t*)stat_rows + 8;
    atomicAdd(dev + 0, (uint32_t)(pl_t1 - pl_t0));
    atomicAdd(dev + 1, (uint32_t)(pl_t2 - pl_t1));
    atomicAdd(dev + 2, (uint32_t)(pl_t3 - pl_t2));
    atomicAdd(dev + 3, (uint32_t)(wall_clock64() - pl_t3));
  }
#endif
}
static __global__ __launch_bounds__(PLAN_SPARSE_MAX_PAIRS) void k_plan_sparse(PlanArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t s_key[PLAN_SPARSE_MAX_PAIRS];
  __shared__ __attribute__((aligned(16))) uint2 s_list[PLAN_SPARSE_MAX_PAIRS];
  __shared__ uint32_t s_xf[9], s_q[PLAN_SQ_WORDS];
  plan_sparse_body<false>(a, s_key, s_xf, s_q, nullptr, nullptr, s_list);
}

// ---- latency front, second half: probe selection of every query + the work list, ONE launch ----------------------
// A single query spent 24 us in k_select_probes (its byte-radix passes put all 4096 keys of the top bytes into one LDS
// bin: distances to centroids share sign and exponent) and 6 us in k_plan_sparse, each behind its own launch.  Here:
// one 1024-thread workgroup per query finishes the coarse scores (k_coarse_lat leaves the raw dot chains; the
// contract's `fma(-2, dot, |q|^2 + |c|^2)` / `1 - dot` is applied here, bit for bit what k_coarse_small writes), keeps
// their sort keys in LDS, radix-selects over the key bits that actually DIFFER (block AND / OR of the keys), emits the
// probe list exactly like k_select_probes (ties at the threshold by ascending partition id, the nearest partition at
// rank 0), and the last workgroup to finish lays out the scan's work list (plan_sparse_body).
#define SELPLAN_NT 1024
#define SELPLAN_MAX_NLIST 16384u  // (the reference's default rows / 8192 partitions: 12 207 at 100 M rows)
struct SelectPlanArgs {
  const float* raw;        // [nq, nlist] dot chains of k_coarse_lat
  const float* qq;         // [nq] |q|^2 chains
  const float* cnorm;      // [nlist]
  uint32_t metric;
  uint32_t nlist, nprobe;
  const uint32_t* plen;
  uint32_t* probes;        // [nq, nprobe]
  unsigned long long* stat_rows;
  uint32_t* qthr;          // [nq] reset to "no bound"
  float* coarse_out;       // [nq, nlist] finished scores (what k_coarse_small would have written), or nullptr
  uint32_t* ticket;        // one word, zero between launches
  PlanArgs plan;
  // select_only (round 6: batches of any size, nlist <= SELPLAN_MAX_NLIST): `raw` holds FINISHED scores (k_coarse_mfma), the workgroup
  // selects its query's probes — the same set, the nearest partition at rank 0 — adds the probed rows to the statistics, and nobody
  // plans: a 1024-thread workgroup that radix-selects over the key bits that DIFFER with its keys in LDS takes ~8 us per query where
  // k_select_probes' four byte passes over global memory took 22 (176 -> ~50 us per 2048 queries at 12 207 partitions).
  uint32_t select_only;
};
static __global__ __launch_bounds__(SELPLAN_NT) void k_select_plan(SelectPlanArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
  uint32_t* s_keys = (uint32_t*)sp_smem;  // [nlist]
  __shared__ uint32_t hist[256];
  __shared__ uint32_t s_and, s_or, s_prefix, s_need, s_less, s_wave_cnt[SELPLAN_NT / 64], s_running, s_best_at, s_eq_all, s_last;
  __shared__ unsigned long long s_best;
  __shared__ __attribute__((aligned(16))) uint32_t s_key[PLAN_SPARSE_MAX_PAIRS];
  __shared__ __attribute__((aligned(16))) uint2 s_list[PLAN_SPARSE_MAX_PAIRS];
  __shared__ uint32_t s_xf[9], s_q[PLAN_SQ_WORDS];
  constexpr int NT = SELPLAN_NT, NW = NT / 64;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const uint32_t b = blockIdx.x, nlist = a.nlist, nprobe = a.nprobe;
  uint32_t* out = a.probes + (size_t)b * nprobe;
  if (a.select_only && !a.plan.act.on(b)) return;  // (an inactive slot of a device-side batch size: like k_select_probes)
#ifdef MI355_DEV_FRONT  // dev: query 0's stage times -> DevCtl::dev[4..7] (keys / radix windows / emit + ticket / plan)
  const unsigned long long sp_t0 = wall_clock64();
#endif
  if (tid == 0) {
    a.qthr[b] = 0xFFFFFFFFu;
    s_and = 0xFFFFFFFFu;
    s_or = 0;
    s_need = nprobe;
    s_less = 0;
    s_running = 0;
    s_best = ~0ull;
    s_best_at = 0;
    s_eq_all = 0;
  }
  __syncthreads();
  {
    const float* src = a.raw + (size_t)b * nlist;
    const float qq = (a.select_only || a.metric == MI355_METRIC_DOT) ? 0.f : a.qq[b];
    uint32_t k_and = 0xFFFFFFFFu, k_or = 0;
    constexpr int KPT = (int)(SELPLAN_MAX_NLIST / SELPLAN_NT);  // scores per thread: all loaded before the first is used
    float s_acc[KPT], s_cn[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const uint32_t pc = min((uint32_t)tid + (uint32_t)i * NT, nlist - 1u);
      s_acc[i] = src[pc];
      s_cn[i] = a.select_only ? 0.f : a.cnorm[pc];
    }
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const uint32_t p = (uint32_t)tid + (uint32_t)i * NT;
      if (p < nlist) {
        const float v = a.select_only ? s_acc[i] : a.metric == MI355_METRIC_DOT ? 1.0f - s_acc[i] : __fmaf_rn(-2.0f, s_acc[i], qq + s_cn[i]);
        if (a.coarse_out) a.coarse_out[(size_t)b * nlist + p] = v;
        const uint32_t key = f32_sort_key(v);
        s_keys[p] = key;
        k_and &= key;
        k_or |= key;
      }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      k_and &= (uint32_t)__shfl_xor((int)k_and, off);
      k_or |= (uint32_t)__shfl_xor((int)k_or, off);
    }
    if (lane == 0) {
      atomicAnd(&s_and, k_and);
      atomicOr(&s_or, k_or);
    }
  }
  __syncthreads();
#ifdef MI355_DEV_FRONT
  const unsigned long long sp_t1 = wall_clock64();
#endif
  const uint32_t diff = s_and ^ s_or;
  // bits above `top` are the same in every key: they are the threshold's too
  uint32_t top = diff ? 32u - (uint32_t)__clz(diff) : 0u;
  uint32_t prefix = top >= 32u ? 0u : (s_and & ~((1u << top) - 1u));
  uint32_t need = nprobe;
  bool eq_all = !diff && nlist == nprobe;  // (no window at all: every key is the threshold)
  while (top > 0u) {  // workgroup-uniform
    const uint32_t w = top < 8u ? top : 8u, shift = top - w;
    const uint32_t himask = top >= 32u ? 0u : ~((1u << top) - 1u);
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (uint32_t p = tid; p < nlist; p += NT) {
      const uint32_t key = s_keys[p];
      if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & ((1u << w) - 1u)], 1u);
    }
    __syncthreads();
    uint32_t h = 0, inc = 0;
    if (tid < 256) {  // inclusive scan of the 256 counts over four waves (whole waves: tid < 256 is wave-uniform)
      h = hist[tid];
      inc = h;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, off);
        if (lane >= off) inc += v;
      }
      if (lane == 63) s_wave_cnt[wid] = inc;
    }
    __syncthreads();
    if (tid < 256) {
      uint32_t base = 0;
      for (int w2 = 0; w2 < wid; ++w2) base += s_wave_cnt[w2];
      inc += base;
      if (inc >= need && inc - h < need) {  // exactly one thread: the bin holding the need-th smallest key
        s_need = need - (inc - h);
        s_prefix = prefix | ((uint32_t)tid << shift);
        s_eq_all = (shift == 0u && h == need - (inc - h)) ? 1u : 0u;
      }
    }
    __syncthreads();
    need = s_need;
    prefix = s_prefix;
    eq_all = s_eq_all != 0u;
    top = shift;
    __syncthreads();  // (s_need / s_prefix are rewritten by the next window)
  }
#ifdef MI355_DEV_FRONT
  const unsigned long long sp_t2 = wall_clock64();
#endif
  const uint32_t T = prefix;
  const uint32_t need_eq = need;             // rows with key == T to take (>= 1)
  const uint32_t n_less = nprobe - need_eq;  // rows with key < T
  // The list is assembled in LDS: the nearest partition moves to rank 0 there, the list goes to global memory once,
  // and a single query's planner reads it where it is — every global round trip this kernel waits for costs ~1.5 us
  // (the first version read its own list back from L2 twice and waited behind three fences: 22 of its 30 us).
  uint32_t* s_out = s_keys + nlist;  // [nprobe]
  if (eq_all) {  // no tie is cut at the threshold: every key <= T, any order
    for (uint32_t p = tid; p < nlist; p += NT) {
      const uint32_t key = s_keys[p];
      if (key <= T) {
        s_out[atomicAdd(&s_less, 1u)] = p;
        atomicMin(&s_best, ((unsigned long long)key << 32) | p);
      }
    }
  } else {
    for (uint32_t p0 = 0; p0 < nlist; p0 += NT) {
      const uint32_t p = p0 + tid;
      const uint32_t key = p < nlist ? s_keys[p] : 0xFFFFFFFFu;
      const bool less = p < nlist && key < T;
      const bool eq = p < nlist && key == T;
      if (less) {
        s_out[atomicAdd(&s_less, 1u)] = p;
        atomicMin(&s_best, ((unsigned long long)key << 32) | p);
      }
      // ordered rank among the equal keys (ascending partition id)
      const uint64_t bal = __ballot(eq);
      if (lane == 0) s_wave_cnt[wid] = (uint32_t)__popcll((unsigned long long)bal);
      __syncthreads();
      uint32_t base = s_running, all = 0;
      for (int w2 = 0; w2 < NW; ++w2) {
        const uint32_t c = s_wave_cnt[w2];
        if (w2 < wid) base += c;
        all += c;
      }
      const uint32_t rank = base + (uint32_t)__popcll((unsigned long long)(bal & ((1ull << lane) - 1ull)));
      if (eq && rank < need_eq) {
        s_out[n_less + rank] = p;
        atomicMin(&s_best, ((unsigned long long)key << 32) | p);
      }
      __syncthreads();
      if (tid == 0) s_running += all;
      __syncthreads();
    }
  }
  __syncthreads();
  if (nprobe > 1) {  // the nearest partition (ties: lowest id) moves to rank 0
    const uint32_t best = (uint32_t)s_best;
    for (uint32_t i = tid; i < nprobe; i += NT)
      if (s_out[i] == best) s_best_at = i;
    __syncthreads();
    if (tid == 0 && s_best_at != 0) {
      s_out[s_best_at] = s_out[0];
      s_out[0] = best;
    }
    __syncthreads();
  }
  for (uint32_t i = tid; i < nprobe; i += NT) out[i] = s_out[i];
  if (a.select_only) {  // the query's probed rows -> statistics; no work list from here
    unsigned long long rows = 0;
    for (uint32_t i = tid; i < nprobe; i += NT) rows += a.plen[s_out[i]];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) rows += (unsigned long long)__shfl_xor((long long)rows, off);
    if (lane == 0 && rows && a.stat_rows) atomicAdd(a.stat_rows, rows);
    return;
  }
  const bool alone = gridDim.x == 1u;  // a single query: this workgroup plans from its own LDS copy, no ticket, no fence
  if (!alone) {
    // this query's probe list is at L2 before the ticket is taken: a RELEASE only (every wave waits for its own stores and writes the
    // L2 back) — the full fence also invalidated this XCD's L2 in every wave, and the planner behind it reads the probes with
    // agent-scope loads anyway (emit + ticket 6.1 -> 4.3 us per call of a batch, tests/tools/front_dev_counters.py)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) {
      const uint32_t t = atomicAdd(a.ticket, 1u);
      s_last = (t == gridDim.x - 1u) ? 1u : 0u;
      if (s_last) atomicExch(a.ticket, 0u);  // ready for the next launch
    }
    __syncthreads();
  }
#ifdef MI355_DEV_FRONT
  const unsigned long long sp_t3 = wall_clock64();
#endif
  if (!alone && !s_last) return;
  plan_sparse_body<true>(a.plan, s_key, s_xf, s_q, alone ? s_out : (const uint32_t*)nullptr, a.stat_rows, s_list);
#ifdef MI355_DEV_FRONT
  __syncthreads();
  if (tid == 0 && a.stat_rows) {  // (stat_rows = &DevCtl::rows_scanned, the first member: the counters follow it)
    uint32_t* dev = (uint32_t*)a.stat_rows + 8;  // rows_scanned(2), short_queries, pad, deadline(2), timed_out, bad_probes, dev[0]
    atomicAdd(dev + 4, (uint32_t)(sp_t1 - sp_t0));
    atomicAdd(dev + 5, (uint32_t)(sp_t2 - sp_t1));
    atomicAdd(dev + 6, (uint32_t)(sp_t3 - sp_t2));
    atomicAdd(dev + 7, (uint32_t)(wall_clock64() - sp_t3));
  }
#endif
}

// ------------------------------------------------------------------- scan ----
// Where entry (code c, column j) of a distance table of M columns lives, in dwords from the table base: `at` always,
// `dup` (SK_NONE: none) for the columns stored twice.
__device__ __forceinline__ void sk_lut_slots(uint32_t c, uint32_t j, uint32_t M, uint32_t& at, uint32_t& dup) {
#ifdef SK_DUAL
  // two slabs of 256-B rows; byte address slab*65536 + c*256 + 4u (u >= 64 spills one row on)
  const uint32_t u = j + 32u;
  at = (u >= 64u ? 16384u : 0u) + c * 64u + u;
  dup = j >= M - 31u ? c * 64u + j - (M - 32u) : SK_NONE;
#else
  at = c * 128u + j + 32u;
  dup = j >= M - 31u ? c * 128u + j - (M - 32u) : SK_NONE;
#endif
}

struct SkewArgs {
  IndexView ix;
  const float* cbT;         // [256][m][dsub]
  const float* qp;          // [nq, dim] preprocessed queries
  const uint32_t* probes;   // [nq * nprobe]
  const SkewItem* items;    // [n_items] partition-major work list
  const uint32_t* q_start;  // [9]
  uint32_t* heads;          // [8 * SK_HEAD_STRIDE]
  uint32_t* qthr;           // [nq] running per-query threshold (f32 sort key)
  uint32_t nprobe, kk;
  RangeFilter range;
  RowFilter filter;
  Cand* cand;               // [nq * nprobe][kk]
  uint32_t* cand_cnt;       // [nq * nprobe] rows an item left in its slots (k_merge_cands reads only those)
  // Latency mode (a batch too small to give every CU a work item): a (query, partition) pair becomes
  // n_slices work items, slice s scanning tile positions [nt*s/n_slices, nt*(s+1)/n_slices) of every
  // stream of the partition; each slice builds the distance table again (12 us) and writes its own kk
  // slots.  > 1: SkewItem::pair carries the slice in its top 8 bits.
  uint32_t n_slices;
  uint32_t dbg;
  DevCtl* ctl;              // deadline / counters of the call
  // SLABBED kernels (SkewShape): ix.m columns per row are scanned as n_slabs slabs of M columns (the template width)
  uint32_t n_slabs;
  uint32_t res_floats;      // LDS floats of the residual: dim, or one slab's M * dsub (SLABBED)
  float2* partial;          // [grid][partial_stride]: per-workgroup partial row sums between slabs (n_slabs > 1)
  uint32_t partial_stride;  // float2 elements per workgroup: (tile positions of the longest unit) * 16 units * 64 lanes
  // IMG kernels: the distance tables of the batch, built by k_lut_images (kernels_lut.h) before this launch —
  // [pair][slab][column block][code][16] f32; a work item copies its image into the LDS table instead of building it
  const float* lut_img;
  unsigned long long* dev_tl;  // (-DMI355_DEV_TIMELINE builds) [grid][SK_TL_WORDS], else nullptr
};

__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}

// plain chunks g = G .. CPT-1 of a tile (steps >= 32)
template <int G, int CPT>
__device__ __forceinline__ void skew_plain_chunks(const uint4 (&cv)[CPT], uint32_t lb, uint32_t pb, float& x,
                                                  float& y) {
  if constexpr (G < CPT) {
    skew_chunk_plain<64 * G>(cv[G], lb, pb, x, y);
    skew_plain_chunks<G + 1, CPT>(cv, lb, pb, x, y);
  }
}

// Queue pop for the persistent scan (thread 0 only).  `q` is the queue this
// workgroup currently drains (its own XCD's first, then the others in ring
// order).  Returns the global item index or SK_NONE when every queue is dry.
__device__ __forceinline__ uint32_t sk_pop_sync(const SkewArgs& a, const uint32_t* s_q, uint32_t& q, uint32_t& tried) {
  if (ctl_expired(a.ctl)) tried = 8;  // QueryExecutionOptions.timeout: stop taking work items
  while (tried < 8) {
    const uint32_t q0 = s_q[q], n = s_q[q + 1] - q0;
    uint32_t* head = a.heads + q * SK_HEAD_STRIDE;
    if (__hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) {
      const uint32_t i = atomicAdd(head, 1u);
      if (i < n) return q0 + i;
    }
    q = (q + 1) & 7u;
    ++tried;
  }
  return SK_NONE;
}

// MULTI: kk > 256 — a work item's rows are selected in passes of SCAN_PASS_ROWS (the codes are
// re-scanned per pass against the table built once; pass p keeps the best rows strictly above the
// last row of pass p-1 in the (distance, rowid) order).  MULTI = false compiles to the single-pass
// kernel unchanged.
// OPT (with MULTI; kk > 128): sixteen waves whose lists hold 192 rows — fewer than the
// SCAN_PASS_ROWS a pass selects.  The winners of a work item are spread over its 16 units, so with
// the workgroup-shared threshold (QSHARE below) a wave needs room for about kk_pass / 16 rows plus
// what arrives between two compactions.  Passes are OPTIMISTIC: kk_pass = min(rows left,
// SCAN_PASS_ROWS), a full list is sorted, cut at the shared threshold (WaveList::prune) and
// refilled; a wave that still cannot make room (the item's best rows crowd into one unit:
// adversarial row order, or hundreds of equal distances) raises `s_ovf`, that pass is discarded
// and it and all later passes of the item select SK_SAFE_PASS rows each, which fit any list.
#define SK_SAFE_PASS 128u
// dev builds (-DMI355_DEV_COUNTERS): thread 0 adds the item's phase times (wall_clock64 ticks) and the
// selection's counters to DevCtl.dev[]: 0 LUT build, 1 scan, 2 merge (ticks), 3 items, 4 rows in the lists
// at the merge, 5 optimistic passes redone, 6 ticks wave 0 waited for the slowest wave, 7 rows admitted
// (the counters are summed per workgroup in registers and flushed ONCE when the workgroup has run out of work: one atomic per
//  phase and item on a single cache line serialises 256 CUs — 250 k atomics per launch at the reference's default shape, where
//  items last 15 us, turned a 2.5 ms scan into 6.4 ms and the phase split into fiction)
#ifdef MI355_DEV_COUNTERS
#define SK_DEV(...) __VA_ARGS__
__device__ __forceinline__ void sk_dev_add(uint32_t& acc, uint32_t v) { acc += v; }
#else
#define SK_DEV(...)
#endif
// dev builds (-DMI355_DEV_TIMELINE, tests/tools/scan_timeline.py): thread 0 of every workgroup stamps wall_clock64() at its start and,
// for its first SK_TL_ITEMS items, at the item's start / table ready / its own wave's streams done / every wave done / merge done, plus
// the item's rows — into SkewArgs::dev_tl [grid][SK_TL_WORDS] (the last launch's timeline; mi355_dev_timeline copies it out)
#ifdef MI355_DEV_KNOBS
#define SK_KNOB(...) __VA_ARGS__
#else
#define SK_KNOB(...)
#endif
#ifndef SK_LAT_SHARE_BEST
#define SK_LAT_SHARE_BEST 1
#endif
#ifndef SK_LAT_QSHARE
#define SK_LAT_QSHARE 0
#endif
#ifndef SK_FAST_SEL
#define SK_FAST_SEL 1
#endif
#define SK_TL_ITEMS 6u
#define SK_TL_WORDS (2u + 6u * SK_TL_ITEMS)
#ifdef MI355_DEV_TIMELINE
#define SK_TL(...) __VA_ARGS__
#else
#define SK_TL(...)
#endif
// SLABBED: the generalised form for the widths the plain kernel is not instantiated for (SkewShape): table column j of
// slab s is sub-quantiser s * M + j of the index (an all-zero column past ix.m), the residual in LDS is one slab's,
// and with n_slabs > 1 a work item walks the slabs: build table s, scan code slab s with every row's accumulator
// starting from the row's partial sum of slabs < s (`partial`, written and read back by the same lane), select in the
// last slab.  Every row sum is still LUT[0] + LUT[1] + ... in j order followed by exact `+ 0.0f` terms.
// TWO: an eight-wave workgroup that shares its CU with a second one (a 32-column table is 64 KiB): one builds its table or
// merges while the other scans.  The second launch-bound figure is waves per SIMD: 4 keeps both at <= 128 VGPRs.
// IMG: the table is not built here — k_lut_images (kernels_lut.h) built every pair's table of the batch with the codebook in
// registers, and the item copies its image (4 * M bytes per code row) into LDS: no residual, no codebook stream.
// LAT: the instantiation sliced batches run (a few queries cut into about one work item per CU, ann_scan_skew_lat.hip): every item
// starts without a bound — all of a query's items run at once — so every wave fills and compacts its list; these kernels compact
// by selection (WaveList::compact_select) and keep ONE copy of the selection code per call site.  The batch kernels, whose items
// inherit the query's bound and hardly ever compact, stay as they were (the extra code costs them registers: C3 -2.5 %).
template <int M, int LR, int NT, bool MULTI, bool OPT = false, bool SLABBED = false, bool TWO = false, bool IMG = false, bool LAT = false>
__global__ __launch_bounds__(NT, TWO ? 4 : NT / 256) void k_scan_skew(SkewArgs a) {
  static_assert(!IMG || !MULTI, "table images are scanned in one selection pass");
  static_assert(!TWO || (NT == 512 && M <= 32), "two workgroups per CU: eight waves and a one-slab table each");
  static_assert(!OPT || (MULTI && LR * 64 >= (int)SK_SAFE_PASS), "OPT rides on the pass machinery");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NW = NT / MI355_WAVE;
  constexpr int P = 128;       // LUT pitch in dwords (sk_pitch_dwords)
  [[maybe_unused]] constexpr int PB = P * 4;  // (SK_ADDR_BFE address form)
  constexpr int CPT = M / 16;  // 1-KiB chunks per tile
  static_assert(M + 32 <= P, "table columns exceed the pitch");
  const IndexView& ix = a.ix;
  const int tid = threadIdx.x, lane = tid & 63;
  // wave-uniform by construction; tell the compiler so that per-wave loops are scalar loops
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lm = sk_phase(lane);
  float* lut = (float*)smem;                                  // [256][P] (dual: two slabs)
  constexpr uint32_t TABLE_BYTES = sk_table_bytes(M);
  float* res = (float*)(smem + TABLE_BYTES);                  // [dim] (SLABBED: [M * dsub], the current slab's)
  const uint32_t res_n = IMG ? 0u : SLABBED ? a.res_floats : ix.dim;  // residual elements of the first (only) slab
  ListEnt* lists = (ListEnt*)(smem + TABLE_BYTES + (((size_t)res_n * 4 + 15) & ~(size_t)15));
  uint32_t* s_cnt = (uint32_t*)(lists + (size_t)NW * LR * MI355_WAVE);  // [NW]
  uint32_t* s_part = s_cnt + NW;                              // [NW] every wave's q-th best (sort key), QSHARE
  uint32_t* s_ovf = s_part + NW;                              // [1] OPT: a list overflowed in the optimistic pass
  uint32_t* s_thr = s_ovf + 1;                                // [1] block threshold (sort key)
  // long candidate lists (kk > 64): the shared threshold is built from every wave's q-th best,
  // q = ceil(kk / NW) (WaveList QTRACK); the kk <= 64 kernel keeps the per-wave bound alone
  constexpr bool QSHARE = (LR >= 3 && !LAT) || (LAT && SK_LAT_QSHARE);  // (LAT: every item of a sliced batch starts without a bound — the shared one is what ends the filling)
  uint32_t* s_q = s_thr + 1;                                  // [9] queue bounds
  // (an OFFSET from the LDS base, not a pointer rounded through size_t: that cast lost the address space and every access to the
  //  records became a FLAT instruction — counted in vmcnt AND lgkmcnt, free to complete out of order with the global loads around
  //  it; 14 of them in a kernel, found while looking for the image-prefetch failures of NOTES 11.11 — which this did not cure)
  const uint32_t rec_off = ((uint32_t)((const unsigned char*)(s_q + 9) - smem) + 31u) & ~31u;  // (smem is 16-B aligned at LDS address 0)
  SkewItem* s_rec = (SkewItem*)(smem + rec_off);  // [2] current / next item
  PassFloor* s_floor = (PassFloor*)(s_rec + 2);                          // [1] (MULTI)
  // the gather address is (code << 9) | column bytes: the table must start at LDS address 0, i.e. the kernel must own
  // no static __shared__ in front of its dynamic block — a compile-time property the launcher checks on the host
  // (launch_scan_skew_m: hipFuncGetAttributes().sharedSizeBytes == 0, else MI355_ERR_NOT_SUPPORTED; nothing traps here)
  const uint32_t lb = 4u * (32u - lm);  // this lane's column origin (bytes)
#ifdef SK_DUAL
  const uint32_t slab_bit = 0x10000u;
#elif defined(SK_ADDR_BFE)
  const uint32_t pb = (uint32_t)__builtin_amdgcn_readfirstlane(PB);        // table pitch
#else
  const uint32_t pb = (uint32_t)__builtin_amdgcn_readfirstlane(0x1fe00);   // code field of the address
#endif
  const uint64_t* rid = ix.row_ids;
  const bool ranged = a.range.has_lower || a.range.has_upper;

  // ---- first item (synchronous) ---------------------------------------------
  uint32_t q_cur = 0, q_tried = 0;  // thread 0's queue cursor
  if (tid < 9) s_q[tid] = a.q_start[tid];
  __syncthreads();
  if (tid == 0) {
    q_cur = xcc_id();
    const uint32_t gi = sk_pop_sync(a, s_q, q_cur, q_tried);
    SkewItem it;
    it.pair = SK_NONE;
    if (gi != SK_NONE) it = a.items[gi];
    s_rec[0] = it;
  }
  __syncthreads();
  // every thread's share of the first residual
  float pre_q[4], pre_c[4];  // up to 4 elements per thread (dim <= 4 * NT, checked at open)
  // IMG: the item's table image instead — quads of four columns, thread e reads quad e, e + NT, ... (coalesced 16-B loads,
  // all of a thread's loads in flight at once); requested where the residual operands are: for the first item before the
  // loop, for every later one while the previous item's lists are merged (SK_IMG_PREFETCH=0: at the item's own start)
  // (round 6, late: with the request in flight across the previous item's merge a search returned rows with wrong distances once in
  //  ~15 searches of 520 x 14 pairs at m = 96 / 192.  The cause was NOT the prefetch — a barrier inside a branch that was not
  //  workgroup-uniform in the block merge, see `tk0` below; the prefetch only widened its window — and with it fixed a prefetching
  //  build is clean (0 of 600 searches).  The prefetch measured worth nothing by itself (NOTES 11.2) and stays off.)
#ifndef SK_IMG_PREFETCH
#define SK_IMG_PREFETCH 0
#endif
#ifndef SK_IMG_NT_LOAD
#define SK_IMG_NT_LOAD 1  // (the image is read once: it should not push the partition codes other queries reuse out of L2; scan -1.5 %)
#endif
  constexpr uint32_t IMG_QPR = (uint32_t)M / 4u, IMG_NQ = 256u * IMG_QPR;  // quads per code row / per table
  constexpr int IMG_PER = IMG ? (int)((IMG_NQ + NT - 1) / NT) : 1;
  sk_f32x4 img_pf[IMG_PER];
  auto fetch_image = [&](uint32_t pair_, uint32_t slab_, sk_f32x4 (&v)[IMG_PER]) {
    const uint32_t n_sl_img = SLABBED ? a.n_slabs : 1u;
    const sk_f32x4* src = (const sk_f32x4*)(a.lut_img + ((size_t)pair_ * n_sl_img + slab_) * (256u * (uint32_t)M));
#pragma unroll
    for (int u = 0; u < IMG_PER; ++u) {
      const sk_f32x4* p = src + sk_min_u32((uint32_t)tid + (uint32_t)u * NT, IMG_NQ - 1u);
      v[u] = SK_IMG_NT_LOAD ? __builtin_nontemporal_load(p) : *p;
    }
  };
  auto prefetch_res = [&](const SkewItem& it) {
    if constexpr (IMG) {
      if (SK_IMG_PREFETCH) fetch_image(a.n_slices > 1u ? (it.pair & 0xFFFFFu) : it.pair, 0u, img_pf);
      return;
    }
    const uint32_t pr = a.n_slices > 1u ? (it.pair & 0xFFFFFu) : it.pair;  // (sliced pairs carry the slice on top: sk_pack_pair)
    const float* q = a.qp + (size_t)(pr / a.nprobe) * ix.dim;
    const float* c = ix.centroids + (size_t)it.part * ix.dim;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t d = tid + u * NT;
      pre_q[u] = 0.f;
      pre_c[u] = 0.f;
      if (d < res_n && d < ix.dim) {
        pre_q[u] = q[d];
        if (ix.metric != MI355_METRIC_DOT) pre_c[u] = c[d];
      }
    }
  };
  if (s_rec[0].pair != SK_NONE) prefetch_res(s_rec[0]);
  // IMG kernels pop TWO items ahead.  The pop is an atomic on a queue head every CU hammers, and the record load depends on
  // it: ~2-3 us of round trips that the 12 us in-item table build used to cover.  With the table phase down to an image copy
  // they would sit on the item's critical path (the workgroup waits for thread 0 at the barrier in front of the scan), so:
  // at the start of item i thread 0 loads the record of item i + 1 from the index it popped at the start of item i - 1 —
  // long since arrived — and pops the index of item i + 2.  (pa, pa_q0, pa_n, pa_q): the older pop, its queue's bounds
  // and id when it was issued.
#ifndef SK_POP2
#define SK_POP2 1
#endif
  constexpr bool POP2 = IMG && SK_POP2;
  uint32_t pa = SK_NONE, pa_q0 = 0, pa_n = 0, pa_q = 0;
  if (POP2 && tid == 0 && s_rec[0].pair != SK_NONE && q_tried < 8) {
    pa_q = q_cur;
    pa_q0 = s_q[q_cur];
    pa_n = s_q[q_cur + 1] - pa_q0;
    pa = atomicAdd(a.heads + q_cur * SK_HEAD_STRIDE, 1u);
  }

  SK_DEV(uint32_t dv_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};)
  SK_TL(uint32_t tl_n = 0; unsigned long long* tl = a.dev_tl ? a.dev_tl + (size_t)blockIdx.x * SK_TL_WORDS : nullptr;
        if (tid == 0 && tl) { tl[0] = wall_clock64(); tl[1] = xcc_id(); })
  for (uint32_t slot = 0;; slot ^= 1u) {
    // the record is wave-uniform: keep it in SGPRs
    const SkewItem* rec = s_rec + slot;
    auto uni32 = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto uni64 = [&](uint64_t v) -> uint64_t { return ((uint64_t)uni32((uint32_t)(v >> 32)) << 32) | (uint64_t)uni32((uint32_t)v); };
    const uint32_t pair_f = uni32(rec->pair);
    if (pair_f == SK_NONE) break;
    const uint32_t pair = a.n_slices > 1u ? (pair_f & 0xFFFFFu) : pair_f;
    const uint32_t slice = a.n_slices > 1u ? ((pair_f >> 20) & 63u) : 0u;
    const uint32_t n_sl = a.n_slices > 1u ? (pair_f >> 26) + 1u : 1u;  // slices THIS pair was cut into (<= a.n_slices, the slot stride)
    const uint32_t oslot = pair * a.n_slices + slice;  // candidate slots / count of this work item
    const uint32_t len = uni32(rec->len);
    const uint32_t lrow0 = uni32(rec->lrow0);
    const uint64_t grow0 = uni64(rec->grow0);
    const uint64_t code_off = uni64(rec->code_off);
    const uint32_t b = pair / a.nprobe;
    const uint32_t n_tiles = (len + SK_TILE - 1) / SK_TILE;
    Cand* out = a.cand + (size_t)oslot * a.kk;
    SK_TL(unsigned long long* tli = (tid == 0 && tl && tl_n < SK_TL_ITEMS) ? tl + 2 + 6 * tl_n : nullptr;
          if (tli) { tli[0] = wall_clock64(); tli[5] = ((unsigned long long)len << 32) | pair_f; })
    SK_DEV(const unsigned long long dv_t0 = wall_clock64(); unsigned long long dv_scan = 0, dv_merge = 0; uint32_t dv_adm = 0;
           const unsigned long long dv_c0 = clock64();)  // shader-clock ticks of the item -> dev[5] (with dev[0..2]: the clock the chip holds)

    // ---- pop the NEXT item now; its index arrives behind the LUT phase's loads
    uint32_t pf = SK_NONE, pf_q0 = 0, pf_n = 0, pf_q = 0;
    if (tid == 0 && q_tried < 8 && ctl_expired(a.ctl)) q_tried = 8;
    if (tid == 0 && q_tried < 8) {
      pf_q = q_cur;
      pf_q0 = s_q[q_cur];
      pf_n = s_q[q_cur + 1] - pf_q0;
      pf = atomicAdd(a.heads + q_cur * SK_HEAD_STRIDE, 1u);
    }
    // POP2: `pf` is the index of the item after the next one; the next item's is the older pop
    const uint32_t nx = POP2 ? pa : pf, nx_q0 = POP2 ? pa_q0 : pf_q0, nx_n = POP2 ? pa_n : pf_n, nx_q = POP2 ? pa_q : pf_q;
    if (POP2) {
      pa = pf;
      pa_q0 = pf_q0;
      pa_n = pf_n;
      pa_q = pf_q;
    }

    // ---- K2: residual (prefetched) + distance table ---------------------------
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t d = tid + u * NT;
      if (d < res_n) res[d] = pre_q[u] - pre_c[u];  // dot: pre_c == 0, q - 0 == q exactly
    }
    if (tid == 0) *s_thr = __hip_atomic_load(a.qthr + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((QSHARE || LAT) && tid < NW) s_part[tid] = 0xFFFFFFFFu;
    if (OPT && tid == 0) *s_ovf = 0u;
    __syncthreads();
    // distance table of slab `slab` (the whole row's when !SLABBED): columns = sub-quantisers slab * M .. of the index
    auto build_lut = [&](uint32_t slab) __attribute__((always_inline)) {
      const uint32_t dsub = ix.dsub;
      const bool dotm = ix.metric == MI355_METRIC_DOT;
      const uint32_t jbase = SLABBED ? slab * (uint32_t)M : 0u;
      const uint32_t n_codes = ix.nbits == 4 ? 16u : 256u;  // table rows that exist (4-bit PQ: codes 0..15, one per byte of the streams)
      auto put = [&](uint32_t e, float acc, bool valid) {
        const uint32_t c = e / (uint32_t)M, j = e % (uint32_t)M;
        if (dotm) acc = 1.0f - acc;
        if (SLABBED && !valid) acc = 0.f;  // a padding column: `+ 0.0f` is exact
        uint32_t at, dup;
        sk_lut_slots(c, j, (uint32_t)M, at, dup);
        lut[at] = acc;
        if (dup != SK_NONE) lut[dup] = acc;
      };
      // entry e = (code c, column j) -> its codebook vector [256][ix.m][dsub] and whether the column exists
      auto cb_of = [&](uint32_t e, bool& valid) -> size_t {
        if constexpr (SLABBED) {
          const uint32_t c = e / (uint32_t)M, jg = jbase + e % (uint32_t)M;
          valid = jg < ix.m;
          return (size_t)c * ix.m + jg;
        } else {
          valid = true;
          return e;
        }
      };
      // sub-vector lengths 4 / 8 / 16 (dim / m of the reference's defaults, index/vector.rs:306-310: 768 / 96,
      // 1536 / 96): whole entries as 16-B loads, 8 of them in flight per thread, the chain in element order
      auto lut_fast_m = [&](auto ds_tag, auto ks_tag, auto dot_tag) {
        constexpr bool dotm = decltype(dot_tag)::value;  // (shadows the run-time flag: no metric branch inside the chains)
        constexpr int DS = decltype(ds_tag)::value;
        constexpr uint32_t KSUB = decltype(ks_tag)::value;  // table rows that exist: 256, or 16 (4-bit PQ)
        constexpr int V = DS / 4;    // 16-B pieces per codebook entry
        constexpr int EPR = (SK_LUT_INFLIGHT / V) > 0 ? (SK_LUT_INFLIGHT / V) : 1;  // entries per thread per round
        constexpr uint32_t TOTAL = KSUB * (uint32_t)M;
        // (ext-vector registers and unconditional loads at a clamped entry: an array of more than eight HIP float4
        //  structs stays in scratch and a load guarded per element is waited for on its own — scripts/check_scratch.py)
        auto load_round = [&](uint32_t e0, sk_f32x4 (&cv4)[EPR][V], bool (&ok)[EPR]) {
#pragma unroll
          for (int u = 0; u < EPR; ++u) {
            const uint32_t e = e0 + u * NT;
            bool valid;
            const size_t at = cb_of(e < TOTAL ? e : TOTAL - 1u, valid);
            ok[u] = valid && e < TOTAL;
            const float* src = a.cbT + (SLABBED && !valid ? (size_t)0 : at) * DS;
#pragma unroll
            for (int v = 0; v < V; ++v) cv4[u][v] = *(const sk_f32x4*)(src + 4 * v);
          }
        };
        auto compute_round = [&](uint32_t e0, const sk_f32x4 (&cv4)[EPR][V], const bool (&ok)[EPR]) {
#pragma unroll
          for (int u = 0; u < EPR; ++u) {
            const uint32_t e = e0 + u * NT;
            if (e < TOTAL) {
              const float* rj = res + (e % (uint32_t)M) * DS;
              float acc = 0.f;
              if (!SLABBED || ok[u]) {
#pragma unroll
                for (int v = 0; v < V; ++v) {
                  const float4 r = *(const float4*)(rj + 4 * v);
                  const sk_f32x4 c = cv4[u][v];
                  if (dotm) {
                    acc = __fmaf_rn(r.x, c.x, acc);
                    acc = __fmaf_rn(r.y, c.y, acc);
                    acc = __fmaf_rn(r.z, c.z, acc);
                    acc = __fmaf_rn(r.w, c.w, acc);
                  } else {
                    const float d0 = r.x - c.x, d1 = r.y - c.y, d2 = r.z - c.z, d3 = r.w - c.w;
                    acc = __fmaf_rn(d0, d0, acc);
                    acc = __fmaf_rn(d1, d1, acc);
                    acc = __fmaf_rn(d2, d2, acc);
                    acc = __fmaf_rn(d3, d3, acc);
                  }
                }
              }
              put(e, acc, ok[u]);
            }
          }
        };
        // (measured and not kept, round 5: a two-set software pipeline of these rounds — 27.7 -> 46.5 ms per C3 launch, the
        //  second register set spills under the 128-VGPR cap; 12 / 16 / 24 pieces in flight per thread instead of 8:
        //  -0.7 / -3.5 / -9 % QPS; entry indices carried incrementally ((code, column) += (NT / M, NT % M), scalar base +
        //  32-bit offsets) instead of e / M, e % M and 64-bit address adds: -1.3 %.  The build streams 786 KB of codebook
        //  per table from L2 and is not bound by its instruction count; profiles/r05_lut_build_ab.txt)
        for (uint32_t e0 = tid; e0 < TOTAL; e0 += EPR * NT) {
          sk_f32x4 cv4[EPR][V];
          bool ok[EPR];
          load_round(e0, cv4, ok);
          compute_round(e0, cv4, ok);
        }
      };
      auto lut_fast = [&](auto ds_tag, auto ks_tag) {
        if (dotm) lut_fast_m(ds_tag, ks_tag, std::true_type{});
        else lut_fast_m(ds_tag, ks_tag, std::false_type{});
      };
      constexpr std::integral_constant<uint32_t, 256u> K256{};
      constexpr std::integral_constant<uint32_t, 16u> K16{};
      if (n_codes == 256u && dsub == 8) {
        lut_fast(std::integral_constant<int, 8>{}, K256);
      } else if (n_codes == 256u && dsub == 16) {
        lut_fast(std::integral_constant<int, 16>{}, K256);
      } else if (n_codes == 256u && dsub == 4) {
        lut_fast(std::integral_constant<int, 4>{}, K256);
      } else if (n_codes == 16u && dsub == 8) {
        lut_fast(std::integral_constant<int, 8>{}, K16);
      } else if (n_codes == 16u && dsub == 16) {
        lut_fast(std::integral_constant<int, 16>{}, K16);
      } else {
        for (uint32_t e = tid; e < n_codes * (uint32_t)M; e += NT) {
          bool valid;
          const size_t at = cb_of(e, valid);
          float acc = 0.f;
          if (valid) {
            const float* cb = a.cbT + at * dsub;
            const float* rj = res + (e % (uint32_t)M) * dsub;
            if (dotm) {
              for (uint32_t t = 0; t < dsub; ++t) acc = __fmaf_rn(rj[t], cb[t], acc);
            } else {
              for (uint32_t t = 0; t < dsub; ++t) {
                float df = rj[t] - cb[t];
                acc = __fmaf_rn(df, df, acc);
              }
            }
          }
          put(e, acc, valid);
        }
      }
    };
    // IMG: the table of (pair, slab) arrives as an image; a quad is stored where sk_lut_slots puts its first entry; the
    // columns stored twice (j >= M - 31) are whole quads because 16 | M - 32 (the quad of j = M - 32 also lands on the
    // unused column u = 0).  The 64 lanes of a store hit 4 of the 8 bank quads (16 codes x the 4 quads of a column block,
    // code rows are 256 B apart): half the LDS store rate, ~0.2 us per table — the price of images that the table kernel
    // writes as contiguous 16 KiB runs.
    auto store_image = [&](const sk_f32x4 (&v)[IMG_PER]) {
#ifdef SK_DUAL
#pragma unroll
      for (int u = 0; u < IMG_PER; ++u) {
        const uint32_t e = (uint32_t)tid + (uint32_t)u * NT;
        if (IMG_NQ % NT == 0 || e < IMG_NQ) {
          // image layout [column block of 16][code][4 quads] (kernels_lut.h): quad e -> (block e / 1024, code, quad e % 4)
          const uint32_t c = (e >> 2) & 255u, j = (e >> 10) * 16u + (e & 3u) * 4u;
          uint32_t at, dup;
          sk_lut_slots(c, j + 3u, (uint32_t)M, at, dup);  // (the quad's LAST column decides whether the quad is stored twice)
          *(sk_f32x4*)(lut + at - 3u) = v[u];
          if (dup != SK_NONE) *(sk_f32x4*)(lut + dup - 3u) = v[u];
        }
      }
#endif
    };
    auto load_lut_image = [&](uint32_t slab) __attribute__((always_inline)) {
      if (SK_IMG_PREFETCH && slab == 0u) {  // (requested before the loop / during the previous item's merge; IMG kernels are single-pass)
        store_image(img_pf);
      } else {
        sk_f32x4 v[IMG_PER];
        fetch_image(pair, slab, v);
        store_image(v);
      }
    };
    auto make_lut = [&](uint32_t slab) __attribute__((always_inline)) {
      if constexpr (IMG) load_lut_image(slab);
      else build_lut(slab);
    };
    if (!(a.dbg & 1u)) make_lut(0);
    // the next item's record: one dependent load, lands during the scan
    SkewItem nxt;
    nxt.pair = SK_NONE;
    bool nxt_valid = false;
    if (tid == 0 && nx != SK_NONE && nx < nx_n) {
      nxt = a.items[nx_q0 + nx];
      nxt_valid = true;
    }
    __syncthreads();
#ifdef SK_IMG_VERIFY2  // dev: the LDS table against the image in memory (non-slabbed IMG kernels): dev[6] after the store, dev[7] after the scan
    [[maybe_unused]] auto verify_table = [&](uint32_t which) {
      if constexpr (IMG && !SLABBED) {
        sk_f32x4 v2[IMG_PER];
        fetch_image(pair, 0u, v2);
        uint32_t nbad = 0;
#pragma unroll
        for (int u = 0; u < IMG_PER; ++u) {
          const uint32_t e = (uint32_t)tid + (uint32_t)u * NT;
          if (IMG_NQ % NT == 0 || e < IMG_NQ) {
            const uint32_t c = (e >> 2) & 255u, j = (e >> 10) * 16u + (e & 3u) * 4u;
            uint32_t at, dup;
            sk_lut_slots(c, j + 3u, (uint32_t)M, at, dup);
            const sk_f32x4 t = *(const sk_f32x4*)(lut + at - 3u);
            nbad += (__float_as_uint(t.x) != __float_as_uint(v2[u].x) || __float_as_uint(t.y) != __float_as_uint(v2[u].y) ||
                     __float_as_uint(t.z) != __float_as_uint(v2[u].z) || __float_as_uint(t.w) != __float_as_uint(v2[u].w)) ? 1u : 0u;
            if (dup != SK_NONE) {
              const sk_f32x4 t2 = *(const sk_f32x4*)(lut + dup - 3u);
              nbad += (__float_as_uint(t2.x) != __float_as_uint(v2[u].x) || __float_as_uint(t2.w) != __float_as_uint(v2[u].w)) ? 1u : 0u;
            }
          }
        }
        if (nbad) atomicAdd(&a.ctl->dev[which], nbad);
      }
    };
    verify_table(6);
    __syncthreads();
#endif

    SK_TL(if (tli) tli[1] = wall_clock64();)
    SK_DEV(const unsigned long long dv_t1 = wall_clock64();
           if (tid == 0) { sk_dev_add(dv_acc[0], (uint32_t)(dv_t1 - dv_t0)); sk_dev_add(dv_acc[3], 1u);
                         })
    // ---- K3 + K4: skewed ADC scan, one stream per wave ----------------------
    auto idof = [&](uint32_t pos) -> uint64_t { return rid ? rid[pos] : grow0 + (pos - lrow0); };
    const uint32_t n_slabs = SLABBED ? a.n_slabs : 1u;
    const size_t slab_bytes = (size_t)sk_part_chunks(n_tiles, CPT) * 1024u;  // one slab of this partition's block
    uint32_t cur_slab = 0;  // the slab whose table (and residual) is in LDS
    const uint32_t thr0_key = *s_thr;  // the query's bound when this item started (valid for every pass)
    uint32_t pass_rows = SCAN_PASS_ROWS;  // (OPT: optimistic passes of SCAN_PASS_ROWS rows, SK_SAFE_PASS after an overflow)
    bool optimistic = OPT;
    bool rec_stored = false;   // thread 0: the next item's record reached s_rec from inside its wave's scan
    bool tail_done = false;    // the next item's residual operands were requested (after its record was popped)
    bool tail_popped = false;  // the pop half already ran in an optimistic pass that was voided
    for (uint32_t pass_base = 0;; pass_base += pass_rows) {
    const uint32_t kk_pass = MULTI ? min(a.kk - pass_base, pass_rows) : a.kk;
    // the item's tail work (next item's record + residual operands) overlaps the merge of the pass that
    // is known to be the last one by its row budget (a voided optimistic pass re-runs; the tail does not)
    const bool last_known = !tail_done && (!MULTI || pass_base + pass_rows >= a.kk);
    bool fl_on = false;
    float fl_d = 0.f;
    uint64_t fl_id = 0;
    if (MULTI && pass_base) {
      fl_on = true;
      fl_d = s_floor->d;
      fl_id = s_floor->id;
    }
    SK_DEV(const unsigned long long dv_p0 = wall_clock64();)
    WaveList<LR, QSHARE || LAT, LAT && SK_FAST_SEL && !OPT> wl;  // (LAT: the q-th best of a compaction feeds the lean bound below)
    const bool whole_kk = pass_base + kk_pass >= a.kk;  // this pass completes the item's kk rows
    uint32_t pub_g = 0xFFFFFFFFu;                        // the tightest bound this lane sent to the query's global word
    const uint32_t q_share = (kk_pass + NW - 1) / NW;
    wl.init(lists + (size_t)wid * LR * MI355_WAVE, kk_pass, q_share);
    float pub_q = __builtin_huge_valf();
    bool q_sorted = false;  // the one early sort of the list happened
    float thr = f32_from_sort_key(thr0_key);
    if (thr0_key == 0xFFFFFFFFu) thr = __builtin_huge_valf();

    // a finished row: tile position tp of stream w, this lane's row
    float published = __builtin_huge_valf();
    auto consume = [&](float acc, uint32_t w, uint32_t tp) __attribute__((always_inline)) {
      const uint32_t row = (w + SK_STREAMS * tp) * SK_TILE + lane;  // w = stream index
      const float d = finalize_dist(acc, ix.metric, ix.m);
      bool ok = row < len && (ranged ? in_range(d, a.range) : d == d);
      if (MULTI && fl_on) {  // strictly above the previous pass's last row
        const bool tie = ok && d == fl_d;
        ok = ok && d >= fl_d;
        if (__any(tie) && tie) ok = idof(lrow0 + row) > fl_id;
      }
      // tightened by the other waves' compactions
      const uint32_t bk = __hip_atomic_load(s_thr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (bk != 0xFFFFFFFFu) thr = fminf(thr, f32_from_sort_key(bk));
      ok = ok && d <= thr;
      if (__any(ok)) {
        if (a.filter.mode != MI355_FILTER_NONE && ok) ok = row_permitted(idof(lrow0 + row), a.filter);
        if constexpr (OPT) {
          // a list shorter than kk_pass cannot shrink by rank: sort it, cut it at the shared threshold
          if (optimistic && wl.cnt + (uint32_t)__popcll((unsigned long long)__ballot(ok)) > (uint32_t)(LR * MI355_WAVE)) {
            wl.compact(lane, idof);
            thr = fminf(thr, wl.t_run);
            const uint32_t bk2 = __hip_atomic_load(s_thr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (bk2 != 0xFFFFFFFFu) thr = fminf(thr, f32_from_sort_key(bk2));
            wl.prune(thr, lane);
            ok = ok && d <= thr;
            if (wl.cnt + (uint32_t)__popcll((unsigned long long)__ballot(ok)) > (uint32_t)(LR * MI355_WAVE)) {
              if (lane == 0) __hip_atomic_store(s_ovf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              ok = false;  // the pass is void: the item is redone in safe passes
            }
          }
        }
        SK_DEV(dv_adm += (uint32_t)__popcll((unsigned long long)__ballot(ok));)
        if constexpr (LAT) {
          // room for this position's rows is made HERE, the one place of a call site where a LAT kernel compacts (a list of
          // 128 entries takes 64 more while it holds up to 64): the compaction's code — a radix select and, for ties across
          // its boundary, the ranking form — exists once per call site, not once per append, early sort and stream end
          static_assert(!LAT || LR * MI355_WAVE >= 2 * MI355_WAVE, "a LAT list holds two appends");
          if (wl.cnt > (uint32_t)((LR - 1) * MI355_WAVE)) {
            wl.compact(lane, idof);
            thr = fminf(thr, wl.t_run);
#if SK_LAT_SHARE_BEST
            if constexpr (!QSHARE) {
              // the waves' BEST rows (see the block merge below): as soon as kk waves have compacted once, the largest of their
              // bests bounds kk rows of the item — an order of magnitude below a wave's own kk-th best.  Any snapshot of
              // s_part is valid: a wave's entry only falls, and the wave keeps a row at or below every value it published.
              // (k <= 16: q = 1, a wave's best is one minimum over its compacted list; beyond: its q-th best, q = ceil(kk / 16), which
              //  the compaction's second radix select left in t_q — waves that hold q rows each bound n * q rows by their largest)
              {
                uint32_t mk = 0xFFFFFFFFu;
                if (q_share == 1u) {
                  if ((uint32_t)lane < wl.cnt) mk = f32_sort_key(wl.list[lane].d);
#pragma unroll
                  for (int off = 1; off < MI355_WAVE; off <<= 1) mk = min(mk, (uint32_t)__shfl_xor((int)mk, off));
                } else if (wl.t_q < __builtin_huge_valf()) {
                  mk = f32_sort_key(wl.t_q);
                }
                if (lane == 0) __hip_atomic_store(s_part + wid, mk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t vp = lane < NW ? __hip_atomic_load(s_part + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0xFFFFFFFFu;
                if ((uint32_t)__popcll((unsigned long long)__ballot(vp != 0xFFFFFFFFu)) * q_share >= kk_pass) {
                  uint32_t v = vp != 0xFFFFFFFFu ? vp : 0u;
#pragma unroll
                  for (int off = 1; off < NW; off <<= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
                  v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
                  if (lane == 0 && atomicMin(s_thr, v) > v && whole_kk) atomicMin(a.qthr + b, v);
                  thr = fminf(thr, f32_from_sort_key(v));
                }
              }
            }
#endif
            ok = ok && d <= thr;
          }
          wl.append_room(ok, d, lrow0 + row, lane);
        } else {
          wl.append(ok, d, lrow0 + row, thr, lane, idof);
        }
        if (wl.t_run < published) {  // a compaction tightened this wave's kk-th best: share it
          published = wl.t_run;
          if (lane == 0) atomicMin(s_thr, f32_sort_key(published));
        }
        if constexpr (QSHARE) {
          // the first time the wave holds a tile's worth of rows (and at least q), sort them once to
          // learn its q-th best; later compactions (list overflow) keep tightening it
          if (!LAT && !q_sorted && wl.cnt >= q_share && wl.cnt >= MI355_WAVE) {
            q_sorted = true;
            wl.compact(lane, idof);
            thr = fminf(thr, wl.t_run);
          }
          if (wl.t_q < pub_q) {
            // every wave's entry is a bound it holds q rows under, at any time: the maximum over a
            // (possibly stale) snapshot of all NW entries bounds the workgroup's NW * q >= kk best rows
            pub_q = wl.t_q;
            if (lane == 0) __hip_atomic_store(s_part + wid, f32_sort_key(pub_q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            uint32_t v = lane < NW ? __hip_atomic_load(s_part + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0u;
#pragma unroll
            for (int off = 1; off < NW; off <<= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
            v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
            if (v != 0xFFFFFFFFu && lane == 0) {
              atomicMin(s_thr, v);
              // kk rows of the QUERY lie at or below it when this pass completes the item's kk (pass_base rows below
              // the floor + NW * q >= kk_pass above it): the other work items of the query — all running at the same
              // time when a single query is cut into slices — pick it up before they merge
              if (whole_kk && v < pub_g) {
                pub_g = v;
                atomicMin(a.qthr + b, v);
              }
            }
          }
        }
      }
    };

    // Both rows of a tile position.  Once a work item holds a bound nearly every position is rejected whole, and that must be
    // cheap: consume() — twice ~45 instructions, two LDS round trips — has no effect unless a lane passes `d <= thr`, so test
    // that first for both rows with the same refreshed bound (a NaN fails the compare like consume()'s `d == d`; the fused
    // form of finalize_dist is the same single rounding: x * 1 - 0, x * 0.5 - 0, x * 1 - (m - 1)).
    const float fd_scale = ix.metric == MI355_METRIC_COSINE ? 0.5f : 1.0f;
    const float fd_bias = ix.metric == MI355_METRIC_DOT ? -(float)(ix.m - 1) : -0.0f;
    [[maybe_unused]] auto consume2 = [&](const sk_f32x2& acc2, uint32_t sa, uint32_t sb, uint32_t tp) __attribute__((always_inline)) {
#ifdef MI355_DEV_KNOBS
      if (a.dbg & 32u) return;  // dev: no selection at all (what the streams alone cost)
#endif
      const uint32_t bk = __hip_atomic_load(s_thr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (bk != 0xFFFFFFFFu) {
        const float t = f32_from_sort_key(bk);
        thr = t < thr ? t : thr;
      }
      const float d0 = __fmaf_rn(acc2.x, fd_scale, fd_bias), d1 = __fmaf_rn(acc2.y, fd_scale, fd_bias);
      if (!__any(d0 <= thr || d1 <= thr)) return;
      if constexpr (LAT) {
#pragma nounroll
        for (uint32_t h = 0; h < 2u; ++h) consume(h ? acc2.y : acc2.x, h ? sb : sa, tp);  // (ONE copy of the selection code per call site)
      } else {
        consume(acc2.x, sa, tp);
        consume(acc2.y, sb, tp);
      }
    };

#ifdef SK_DUAL
    for (uint32_t slab = 0; slab < n_slabs; ++slab) {
    const bool first_slab = !SLABBED || slab == 0, last_slab = !SLABBED || slab + 1 == n_slabs;
    if constexpr (SLABBED) {
      if (cur_slab != slab) {  // (workgroup-uniform) the table of another slab: also a later pass coming back to slab 0
        __syncthreads();       // every wave is done with the old table and residual
        for (uint32_t dl = tid; !IMG && dl < a.res_floats; dl += NT) {
          const uint32_t dg = slab * a.res_floats + dl;  // res_floats = M * dsub whenever there are several slabs
          float v = 0.f;
          if (dg < ix.dim) {
            v = a.qp[(size_t)b * ix.dim + dg];
            if (ix.metric != MI355_METRIC_DOT) v -= ix.centroids[(size_t)uni32(rec->part) * ix.dim + dg];
          }
          res[dl] = v;
        }
        __syncthreads();
        if (!(a.dbg & 1u)) make_lut(slab);
        __syncthreads();
        cur_slab = slab;
      }
    }
    const uint8_t* pcodes = ix.codes + code_off + (SLABBED ? slab * slab_bytes : (size_t)0);
    // two streams (chains A = 2u, B = 2u + 1) per wave; codes arrive chunk by chunk through a
    // ring of RING register slots per chain (prefetch distance RING - 1 chunks)
    for (uint32_t u = wid; u < SK_UNITS && !(a.dbg & 2u); u += NW) {
      const uint32_t nt = sk_unit_tiles(n_tiles, u);
      if (!nt) continue;
      constexpr int RING = SK_RING_FULL ? CPT : ((CPT % 3 == 0) ? 3 : (CPT % 2 == 0 ? 2 : CPT));
      const uint32_t n_chunks = nt * CPT + SK_TAIL_CHUNKS;  // dual chunks of the unit
      const uint4* src = (const uint4*)(pcodes + (size_t)sk_unit_chunk0(n_tiles, u, CPT) * 1024u) + lane;
      sk_u32x4 ra[RING], rb[RING];
      auto fetch = [&](int slot, uint32_t c) {  // dual chunk c -> ring slot (clamped: the load count per chunk is fixed)
        c = sk_min_u32(c, n_chunks - 1);
        sk_load2(ra[slot], rb[slot], src + (size_t)c * 128);
      };
      // this work item's tile positions of the unit (the whole unit unless the pair is sliced).  A slice that
      // starts inside the stream meets the tails of tile n0 - 1 in its first steps (they only feed Y, which the
      // n > n0 test below never consumes) and ends like the stream does: the first two chunks of position n1
      // hold the tails of tile n1 - 1 (their tile-n1 bytes go to a dummy accumulator)
      const uint32_t n0 = (uint32_t)((uint64_t)nt * slice / n_sl), n1 = (uint32_t)((uint64_t)nt * (slice + 1u) / n_sl);
      if (n0 == n1) continue;
      // partial row sums of the slabs before this one: [tile position][unit][lane] float2 (chains A, B), parked by this
      // very lane in the previous slab.  The load of position n + 1 is issued at the top of position n — before that
      // position's CPT >= RING - 1 ring fetches, so the counted wait that opens position n + 1 has retired it
      // (vmcnt retires in order) — and x starts from it.
      sk_f32x2 px = {0.f, 0.f};
      float2* ppart = nullptr;
      if constexpr (SLABBED) {
        if (n_slabs > 1u) ppart = a.partial + (size_t)blockIdx.x * a.partial_stride + (size_t)u * MI355_WAVE + lane;
        if (!first_slab)
          asm volatile("global_load_dwordx2 %0, %1, off" : "+v"(px) : "v"(ppart + (size_t)n0 * (SK_UNITS * MI355_WAVE)) : "memory");
      }
      auto park = [&](const sk_f32x2& v, uint32_t tp) {  // (not the last slab) this lane's two rows of position tp
        ppart[(size_t)tp * (SK_UNITS * MI355_WAVE)] = make_float2(v.x, v.y);
      };
#ifdef MI355_DEV_SCANSPLIT  // dev: where wave 0's scan phase goes — first chunk's latency / the positions / the tail (dev[4], [6], [7])
      const unsigned long long ss_t0 = wall_clock64();
#endif
#pragma unroll
      for (int g = 0; g < RING; ++g) fetch(g, n0 * CPT + g);
#ifdef MI355_DEV_SCANSPLIT
      sk_wait_codes<2 * (RING - 1)>(ra[0], rb[0]);
      const unsigned long long ss_t1 = wall_clock64();
#endif
      sk_f32x2 x = {0.f, 0.f}, y = {0.f, 0.f};
      uint32_t r = lb, r2 = lb;  // [bit 16: slab][byte 1: code][byte 0: column origin]; r2: chain B's copy (nreg=2)
      const uint32_t sa = u * SK_CHAINS, sb = sa + 1;
      for (uint32_t n = n0; n < n1; ++n) {
        const uint32_t c0 = n * CPT;
        // The SIMD's arbiter favours its oldest wave: left alone, the four waves of a SIMD finish their streams one after
        // the other and the last one runs ~16 us of a 52 us scan with idle pipes around it.  They take turns at the issue
        // priority instead, tile position by tile position (wid + wid / 4 differs between the waves of a SIMD whichever
        // way waves map to SIMDs): C3-shaped scan 14.96 -> 14.26 ms per 1024 queries (profiles/r04_j_*).
        sk_rotate_prio((uint32_t)(wid + (wid >> 2)) + n);
        if constexpr (SLABBED) {
          if (!first_slab) {
            // ONE statement: wait for this position's partial sums, start x from them, request the next position's into
            // the same register (clamped: one load per position, like the ring's).  px is never read by compiler-made code
            // while a load into it is in flight; scripts/check_inflight_regs.py checks the generated code for copies.
            const uint32_t nn = sk_min_u32(n + 1u, n1 - 1u);
            asm volatile("s_waitcnt vmcnt(%c3)\n\tv_mov_b64 %0, %1\n\tglobal_load_dwordx2 %1, %2, off"
                         : "=&v"(x), "+v"(px)
                         : "v"(ppart + (size_t)nn * (SK_UNITS * MI355_WAVE)), "i"(2 * (RING - 1))
                         : "memory");
          }
        }
        auto chunks = [&](auto self, auto gtag) -> void {
          constexpr int G = decltype(gtag)::value;
          if constexpr (G < CPT) {
            sk_wait_codes<2 * (RING - 1)>(ra[G % RING], rb[G % RING]);  // RING-1 younger chunks stay in flight
            skew_dchunk<G>(ra[G % RING], rb[G % RING], r, r2, slab_bit, x, y);
            fetch(G % RING, c0 + G + RING);
            if constexpr (G == 1) {
              if (n > n0) {  // rows of tile position n-1 are complete on every lane after step 30
                if (last_slab) {
                  consume2(y, sa, sb, n - 1);
                } else {
                  park(y, n - 1);
                }
              }
              if (n == n0 && tid == 0 && nxt_valid) {
                s_rec[slot ^ 1u] = nxt;
                rec_stored = true;
              }
            }
            self(self, std::integral_constant<int, G + 1>{});
          }
        };
        chunks(chunks, std::integral_constant<int, 0>{});
        y = x;
        x = sk_f32x2{0.f, 0.f};
        r &= 0xffffu;  // every lane is back in slab 0 at step 0
        r2 &= 0xffffu;
      }
#ifdef MI355_DEV_SCANSPLIT
      const unsigned long long ss_t2 = wall_clock64();
#endif
      {  // 31 more steps finish the last tile's rows (CPT % RING == 0: the tail sits in slots 0, 1)
        sk_f32x2 dummy = {0.f, 0.f};
        sk_wait_codes<0>(ra[0], rb[0]);  // also drains the clamped prefetches: the ring registers die here
        sk_wait_codes<0>(ra[1 % RING], rb[1 % RING]);
        if constexpr (SLABBED) asm volatile("" : "+v"(px));  // (its last, clamped load has landed too: vmcnt(0) above)
        skew_dchunk<0>(ra[0], rb[0], r, r2, slab_bit, dummy, y);
        skew_dchunk<1>(ra[1 % RING], rb[1 % RING], r, r2, slab_bit, dummy, y);
        if (last_slab) {
          consume2(y, sa, sb, n1 - 1);
        } else {
          park(y, n1 - 1);
        }
      }
#ifdef MI355_DEV_SCANSPLIT
      if (tid == 0) {
        const unsigned long long ss_t3 = wall_clock64();
        sk_dev_add(dv_acc[4], (uint32_t)(ss_t1 - ss_t0));
        sk_dev_add(dv_acc[6], (uint32_t)(ss_t2 - ss_t1));
        sk_dev_add(dv_acc[7], (uint32_t)(ss_t3 - ss_t2));
      }
#endif
    }
    }  // slabs
#else
    static_assert(!SLABBED, "slabs ride on the two-rows-per-lane blocks");
    for (uint32_t w = wid; w < SK_STREAMS && !(a.dbg & 2u); w += NW) {
      const uint32_t nt = sk_unit_tiles(n_tiles, w);
      if (!nt) continue;
      const uint4* src = (const uint4*)(pcodes + (size_t)sk_unit_chunk0(n_tiles, w, CPT) * 1024u) + lane;
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
        if (n == 0 && tid == 0 && nxt_valid) {  // landed long ago; frees its registers
          s_rec[slot ^ 1u] = nxt;
          rec_stored = true;
        }
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

#endif

    __builtin_amdgcn_s_setprio(0);
    SK_TL(if (tli) tli[2] = wall_clock64();)
#ifdef SK_IMG_VERIFY2
    __syncthreads();
    verify_table(7);
    __syncthreads();
#endif
    SK_DEV(const unsigned long long dv_pw = wall_clock64();)  // this wave's streams are done
    // ---- block result: exact kk_pass best of all waves' lists, written sorted ----
    if (!LAT && wl.cnt > kk_pass) wl.compact(lane, idof);  // (LAT: the block below shrinks long lists — once, for the workgroup)
    if (lane == 0) s_cnt[wid] = wl.cnt;
    auto next_item_fallback = [&]() {  // thread 0: the prefetch ran off the end of its queue (rare, synchronous)
      if (!nxt_valid) {
        if (POP2 && pa != SK_NONE && pa < pa_n) {
          // the YOUNGER pop holds a work item (it was taken from the queue the cursor moved on to, after the older one had run
          // off the previous queue): it is the next item — nothing popped is ever dropped
          nxt = a.items[pa_q0 + pa];
          pa = SK_NONE;
        } else {
          // (a pop that ran off its queue's end leaves that queue — unless the workgroup already left it: the two-deep pop's
          //  second index of a dry queue arrives after the first one moved the cursor on)
          if (q_tried < 8 && nx_q == q_cur && nx != SK_NONE) {
            q_cur = (q_cur + 1) & 7u;
            ++q_tried;
          }
          const uint32_t gi = sk_pop_sync(a, s_q, q_cur, q_tried);
          nxt.pair = SK_NONE;
          if (gi != SK_NONE) nxt = a.items[gi];
        }
      }
      s_rec[slot ^ 1u] = nxt;
    };
    if (last_known && !tail_popped && tid == 0 && (!nxt_valid || !rec_stored)) next_item_fallback();
    // what the query's other work items learnt meanwhile (a bound on its kk-th best distance: valid for any pass)
    if (tid == 0) {
      const uint32_t g = __hip_atomic_load(a.qthr + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (g != 0xFFFFFFFFu) atomicMin(s_thr, g);
    }
    __syncthreads();
    SK_TL(if (tli) tli[3] = wall_clock64();)
    SK_DEV(const unsigned long long dv_p1 = wall_clock64(); dv_scan += dv_p1 - dv_p0;  // every wave is done
           )
    if constexpr (OPT) {
      if (optimistic && *s_ovf) {  // (workgroup-uniform after the barrier)
        // this pass is void: redo it, and run whatever follows, in passes that fit any list; the floor
        // it started from is untouched (s_floor is only written by a pass's merge)
        optimistic = false;
        tail_popped = tail_popped || last_known;
        SK_DEV(if (tid == 0) sk_dev_add(dv_acc[5], 1u);)
        __syncthreads();  // every thread has read the flag
        if (tid == 0) {
          *s_ovf = 0u;
          *s_thr = thr0_key;
        }
        if (tid < NW) s_part[tid] = 0xFFFFFFFFu;
        __syncthreads();
        pass_rows = SK_SAFE_PASS;
        pass_base -= pass_rows;  // the loop increment brings it back to this pass's base
        continue;
      }
    }
    // the next item's residual operands travel while this item's lists are merged
    if (last_known) {
      if (s_rec[slot ^ 1u].pair != SK_NONE) prefetch_res(s_rec[slot ^ 1u]);
      tail_done = true;
    }
    {
      // The lists were filled under the bounds known when their rows arrived; the bound the workgroup holds now — its
      // own or one of the query's other work items' — cuts them before they are ranked.  (A single query's 512 slices
      // run at the same time without any bound: each would rank, and hand the final merge, its own kk best rows.)
      // The barrier behind this read is UNCONDITIONAL.  Until late in round 6 it sat inside `if (tk0 != no bound)`: with no bound
      // yet — a query's first item on a partition so short that no list filled: 2048 rows are exactly 128 per wave — a fast wave went
      // on to the block below, compacted its list and published its kk-th best to s_thr BEFORE a slow wave had read tk0; the slow wave
      // then saw a bound, took the branch and waited at a barrier the others never reached: from there the waves of the workgroup
      // were one barrier apart (a wave could start the next item while others still merged this one).  Wrong distances once in ~15
      // searches of 520 x 14 pairs with the table-image prefetch widening the window (NOTES 11.11), never caught without it.
      const uint32_t tk0 = *s_thr;
      if (tk0 != 0xFFFFFFFFu) {
        wl.filter(f32_from_sort_key(tk0), lane);
        if (lane == 0) s_cnt[wid] = wl.cnt;
      }
      __syncthreads();
    }
    // Exact (distance, rowid) ranks of the lists' rows, one row per WAVE at a time, the 64 lanes
    // comparing it with 64 rows of the concatenated lists per step: every step is an independent
    // LDS read per lane and one ballot.  (A row per THREAD walking all the lists serially is a chain of
    // dependent LDS reads run by the few lanes whose slots are filled: 13 us of a 100 us item at
    // kk = 250 with 124 rows in the lists, 600 us for an item whose lists are full.)
    // Long lists first shrink to what can still win: an item that ran without a query bound (every
    // query's first one — under best_first its nearest partition) ends with up to LR * 64 rows per wave,
    // admitted under thresholds that were looser than the final one.  Every wave sorts its own list
    // (exact kk-th / q-th best of the wave), the workgroup bound is rebuilt from those, and each list is
    // cut at it: the ranking below then sees ~kk rows instead of ~NW * LR * 64 (its cost is quadratic).
    {
      uint32_t tot0 = 0;
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) tot0 += s_cnt[w2];
      // LAT, first: the bound of the waves' BEST rows needs no compaction — a wave's best is one minimum over its list, whatever
      // order and length (a single-position item arrives here with 128 rows per wave and no bound at all: compacting the 16 lists
      // first was most of a 9 us merge phase at the reference's default shape).  When kk waves hold a row, cut every list at the
      // largest of their bests and count again; lists that stay long take the general route below.
      if constexpr (LAT && !QSHARE) {
        if (tot0 > kk_pass + 2u * MI355_WAVE && q_share == 1u) {  // workgroup-uniform (kk <= 16)
          uint32_t mk = 0xFFFFFFFFu;
#pragma unroll
          for (int r = 0; r < LR; ++r) {
            const uint32_t slot = (uint32_t)(r * MI355_WAVE + lane);
            if (slot < wl.cnt) mk = min(mk, f32_sort_key(wl.list[slot].d));
          }
#pragma unroll
          for (int off = 1; off < MI355_WAVE; off <<= 1) mk = min(mk, (uint32_t)__shfl_xor((int)mk, off));
          if (lane == 0) s_part[wid] = mk;
          __syncthreads();
          const uint32_t vp = lane < NW ? s_part[lane] : 0xFFFFFFFFu;
          const bool bounded = (uint32_t)__popcll((unsigned long long)__ballot(vp != 0xFFFFFFFFu)) >= kk_pass;  // (the same in every wave)
          if (bounded) {
            uint32_t v = vp != 0xFFFFFFFFu ? vp : 0u;
#pragma unroll
            for (int off = 1; off < NW; off <<= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
            v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
            if (tid == 0) {
              atomicMin(s_thr, v);
              if (whole_kk) atomicMin(a.qthr + b, v);
            }
            wl.filter(f32_from_sort_key(v), lane);
            if (lane == 0) s_cnt[wid] = wl.cnt;
          }
          __syncthreads();
          if (bounded) {
            tot0 = 0;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) tot0 += s_cnt[w2];
          }
        }
      }
      if (tot0 > kk_pass + 2u * MI355_WAVE) {  // workgroup-uniform
        wl.compact(lane, idof);
        if (wl.t_run < published) {
          published = wl.t_run;
          if (lane == 0) atomicMin(s_thr, f32_sort_key(published));
        }
        if constexpr (QSHARE) {
          if (lane == 0 && wl.t_q < pub_q) __hip_atomic_store(s_part + wid, f32_sort_key(wl.t_q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // LAT: every wave's BEST row.  Waves that hold a row hold one at or below their own best, so the LARGEST of kk waves' bests
        // bounds kk rows — with 16 waves and k = 10 about the 16th best row of the item, where a
        // wave's own kk-th best (published above) is the 10th best of a sixteenth of it.  The lists then enter the ranking below
        // with a dozen rows in all instead of 160.
        if constexpr (LAT && !QSHARE) {
          uint32_t mk = 0xFFFFFFFFu;
          if (q_share == 1u) {
            if ((uint32_t)lane < wl.cnt) mk = f32_sort_key(wl.list[lane].d);  // (compacted: cnt <= kk_pass <= 16)
#pragma unroll
            for (int off = 1; off < MI355_WAVE; off <<= 1) mk = min(mk, (uint32_t)__shfl_xor((int)mk, off));
          } else if (wl.t_q < __builtin_huge_valf()) {
            mk = f32_sort_key(wl.t_q);  // (the wave's q-th best, q = ceil(kk / 16): left by the compaction above)
          }
          if (lane == 0) s_part[wid] = mk;  // (reset to "no row" at the item's start)
        }
        __syncthreads();
        if constexpr (LAT && !QSHARE) {
          const uint32_t vp = lane < NW ? s_part[lane] : 0xFFFFFFFFu;
          const uint32_t n_have = (uint32_t)__popcll((unsigned long long)__ballot(vp != 0xFFFFFFFFu));
          if (n_have * q_share >= kk_pass) {  // (wave-uniform, the same in every wave)
            // the kk_pass-th smallest of the waves' bests would do; their maximum is one reduction
            uint32_t v = vp != 0xFFFFFFFFu ? vp : 0u;
#pragma unroll
            for (int off = 1; off < NW; off <<= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
            v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
            if (tid == 0) {
              atomicMin(s_thr, v);
              if (whole_kk) atomicMin(a.qthr + b, v);
            }
            __syncthreads();
          }
        }
        if constexpr (QSHARE) {
          if (tid < MI355_WAVE) {  // wave 0: the maximum of the waves' q-th bests bounds NW * q >= kk rows
            uint32_t v = lane < NW ? s_part[lane] : 0u;
#pragma unroll
            for (int off = 1; off < NW; off <<= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
            if (lane == 0 && v != 0xFFFFFFFFu) {
              atomicMin(s_thr, v);
              if (whole_kk) atomicMin(a.qthr + b, v);
            }
          }
          __syncthreads();
        }
        const uint32_t tk = *s_thr;
        if (tk != 0xFFFFFFFFu) {
          if (wl.fast) wl.filter(f32_from_sort_key(tk), lane);  // (a list compacted by selection is not sorted)
          else wl.prune(f32_from_sort_key(tk), lane);
        }
        if (lane == 0) s_cnt[wid] = wl.cnt;
        __syncthreads();
      }
    }
    // LAT, k beyond a few dozen: the bounds above leave ~2.5 kk rows (16 waves x their q-th bests), and the ranking below is
    // quadratic — 50 us of merge per item at k = 100 on single-position items (the reference's default shape).  The EXACT kk-th
    // smallest key of all lists (four histogram passes over the lists where they lie; the dead table's first KiB holds the
    // histogram) cuts them to kk rows plus ties first.
    if constexpr (LAT && !QSHARE) {
      uint32_t tot1 = 0;
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) tot1 += s_cnt[w2];
      if (tot1 > 2u * kk_pass + 32u) {  // workgroup-uniform
        uint32_t* hist = (uint32_t*)smem;
        const uint32_t T = block_kth_smallest_key<NT>(
            [&](uint32_t i) -> uint32_t {
              const uint32_t w2 = i / (uint32_t)(LR * MI355_WAVE), sl = i % (uint32_t)(LR * MI355_WAVE);
              return sl < s_cnt[w2] ? f32_sort_key(lists[i].d) : 0xFFFFFFFFu;
            },
            (uint32_t)(NW * LR * MI355_WAVE), kk_pass, hist, hist + 256);
        wl.filter(f32_from_sort_key(T), lane);
        __syncthreads();  // (every thread read the old counts)
        if (lane == 0) s_cnt[wid] = wl.cnt;
        __syncthreads();
      }
    }
    SK_DEV(const unsigned long long dv_m0 = wall_clock64();)
    uint32_t pre[NW + 1];  // wave-uniform prefix sums of the list lengths
    pre[0] = 0;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) pre[w2 + 1] = pre[w2] + (uint32_t)__builtin_amdgcn_readfirstlane((int)s_cnt[w2]);
    const uint32_t total = pre[NW];
    const uint32_t n_out = min(total, kk_pass);
#ifndef MI355_DEV_SCANSPLIT
    SK_DEV(if (tid == 0) sk_dev_add(dv_acc[4], total);)
#endif
    auto locate = [&](uint32_t c, uint32_t& w2, uint32_t& j2) {  // flat position -> (list, entry)
      w2 = 0;
#pragma unroll
      for (int x = 1; x < NW; ++x) w2 += (c >= pre[x]) ? 1u : 0u;
      uint32_t base = 0;
#pragma unroll
      for (int x = 1; x < NW; ++x) base = (c >= pre[x]) ? pre[x] : base;
      j2 = c - base;
    };
    // ranks are wave-uniform; lane i of the wave parks the i-th row it ranked and the rows are written
    // 64 at a time, so the row-id loads of the output records (one random 8-B read each) run in parallel
    uint32_t parked = 0, r_rank = 0xFFFFFFFFu;
    ListEnt r_ent;
    r_ent.d = 0.f;
    r_ent.pos = 0;
    auto flush = [&]() {
      if ((uint32_t)lane < parked && r_rank < kk_pass) {
        Cand o;
        o.d = r_ent.d;
        o.pos = r_ent.pos;
        o.id = idof(r_ent.pos);
        out[pass_base + r_rank] = o;
        // kk rows at or below this distance exist: a bound for every other partition of this query
        if (pass_base + r_rank == a.kk - 1) atomicMin(a.qthr + b, f32_sort_key(o.d));
        if (MULTI && r_rank == kk_pass - 1) {  // the next pass starts strictly above this row
          s_floor->d = o.d;
          s_floor->id = o.id;
          s_floor->on = 1;
        }
      }
      parked = 0;
    };
    for (uint32_t e = (uint32_t)wid; e < total; e += NW) {  // wave-uniform
      uint32_t ew, ej;
      locate(e, ew, ej);
      const ListEnt mine = lists[(size_t)ew * LR * MI355_WAVE + ej];  // broadcast
      uint32_t rank = 0;
      uint64_t mine_id = 0;
      bool have_id = false;
      for (uint32_t c0 = 0; c0 < total && rank < kk_pass; c0 += MI355_WAVE) {
        const uint32_t c = c0 + lane;
        bool lt = false, tie = false;
        ListEnt o;
        o.d = 0.f;
        o.pos = 0;
        if (c < total) {
          uint32_t w2, j2;
          locate(c, w2, j2);
          o = lists[(size_t)w2 * LR * MI355_WAVE + j2];
          lt = o.d < mine.d;
          tie = o.d == mine.d && o.pos != mine.pos;
        }
        if (__any(tie)) {  // equal distances: the row id decides (fetched only here)
          if (!have_id) {
            mine_id = idof(mine.pos);
            have_id = true;
          }
          if (tie) lt = idof(o.pos) < mine_id;
        }
        rank += (uint32_t)__popcll((unsigned long long)__ballot(lt));
      }
      if ((uint32_t)lane == parked) {
        r_rank = rank;
        r_ent = mine;
      }
      if (++parked == MI355_WAVE) flush();
    }
    SK_DEV(const unsigned long long dv_m1 = wall_clock64();)
    if (parked) flush();
#ifndef MI355_DEV_SCANSPLIT
    SK_DEV(if (tid == 0) { sk_dev_add(dv_acc[6], (uint32_t)(dv_m0 - dv_p1)); sk_dev_add(dv_acc[7], (uint32_t)(dv_m1 - dv_m0)); })
#endif
    const bool more = MULTI && total >= kk_pass && pass_base + kk_pass < a.kk;
    SK_DEV(__syncthreads(); dv_merge += wall_clock64() - dv_p1;)
    if (!more) {
      // the item's slots hold this many rows, ranks 0 .. in (distance, rowid) order; the rest is not written
      if (tid == 0) a.cand_cnt[oslot] = pass_base + n_out;
      break;
    }
    __syncthreads();  // the floor is published; the lists and the block threshold are rebuilt
    if (tid == 0) *s_thr = thr0_key;
    if ((QSHARE || LAT) && tid < NW) s_part[tid] = 0xFFFFFFFFu;
    __syncthreads();
    }  // passes
    if (MULTI && !tail_done) {  // the tail work of an item whose last pass was not known in advance
      if (tid == 0 && (!nxt_valid || !rec_stored)) {
        if (!nxt_valid) {
          if (q_tried < 8 && nx_q == q_cur && nx != SK_NONE) {
            q_cur = (q_cur + 1) & 7u;
            ++q_tried;
          }
          const uint32_t gi = sk_pop_sync(a, s_q, q_cur, q_tried);
          nxt.pair = SK_NONE;
          if (gi != SK_NONE) nxt = a.items[gi];
        }
        s_rec[slot ^ 1u] = nxt;
      }
      __syncthreads();
      if (s_rec[slot ^ 1u].pair != SK_NONE) prefetch_res(s_rec[slot ^ 1u]);
    }
    SK_TL(if (tli) { tli[4] = wall_clock64(); ++tl_n; tl[1] = xcc_id() | ((unsigned long long)tl_n << 8); })
    SK_DEV(if (tid == 0 && !OPT) { sk_dev_add(dv_acc[5], (uint32_t)(clock64() - dv_c0)); }
           if (tid == 0) { sk_dev_add(dv_acc[1], (uint32_t)dv_scan); sk_dev_add(dv_acc[2], (uint32_t)dv_merge); }
           (void)dv_adm;)
    __syncthreads();  // LDS is rebuilt by the next item
  }
  SK_DEV(if (tid == 0) { for (int x = 0; x < 8; ++x) if (dv_acc[x]) atomicAdd(&a.ctl->dev[x], dv_acc[x]); })
}
