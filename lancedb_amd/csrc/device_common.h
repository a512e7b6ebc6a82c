// device_common.h — shared device helpers for the gfx950 ANN kernels.
//
// Arithmetic contract (oracle/ann_oracle.c header): every distance
// is a d-ascending f32 fma chain; this TU is compiled with -ffp-contract=off so
// only the explicit __fmaf_rn calls fuse.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mi355_ann.h"

#define MI355_WAVE 64

struct Cand {  // 16 B candidate record kept between scan -> merge -> refine
  float d;       // final (metric-adjusted) distance
  uint32_t pos;  // local row position on this handle; 0xFFFFFFFF = empty slot
  uint64_t id;   // _rowid
};

#define CAND_EMPTY_POS 0xFFFFFFFFu

// Device-side control word of an index handle (one 64-B line, zeroed per call).
// `deadline` is QueryExecutionOptions.timeout (rust/lancedb/src/query.rs:626-658) as a
// device clock value: the persistent scan polls it between work items, every other kernel
// at entry; the first kernel that sees it pass latches `timed_out` and the rest of the
// call's launch sequence exits at once.
struct DevCtl {
  // per-call counters, zeroed together (DEVCTL_COUNTER_BYTES) unless the profile mode is cumulative
  unsigned long long rows_scanned;  // stats: vectors scanned (sum over (query, partition) pairs)
  uint32_t short_queries;           // queries re-searched over maximum_nprobes partitions (decided on the device)
  uint32_t pad0;
  unsigned long long deadline;      // wall_clock64() value after which kernels stop; 0 = none
  uint32_t timed_out;
  uint32_t bad_probes;              // probe ids outside 0..nlist-1 (mi355_search_probes)
  uint32_t dev[8];                  // -DMI355_DEV_COUNTERS builds only: per-phase ticks / selection counters of the scan
};
#define DEVCTL_COUNTER_BYTES 16u

__device__ __forceinline__ bool ctl_expired(DevCtl* ctl) {
  if (!ctl) return false;
  if (__hip_atomic_load(&ctl->timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
  const unsigned long long dl = ctl->deadline;
  if (dl && (unsigned long long)wall_clock64() > dl) {
    __hip_atomic_store(&ctl->timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
  }
  return false;
}

// first kernel of a call with a timeout: deadline = now + ticks of the constant-rate device clock
// (`reset_counters`: also zero the per-call counters — one launch instead of a memset and a launch)
static __global__ void k_arm_deadline(DevCtl* ctl, unsigned long long ticks, uint32_t reset_counters = 0) {
  ctl->deadline = ticks ? (unsigned long long)wall_clock64() + ticks : 0ull;
  ctl->timed_out = 0;
  if (reset_counters) {
    ctl->rows_scanned = 0ull;
    ctl->short_queries = 0u;
  }
}

// Correctly rounded sqrt / divide.  NOT __fsqrt_rn / __fdiv_rn: without
// OCML_BASIC_ROUNDED_OPERATIONS hipcc maps __fsqrt_rn to the approximate native
// sqrt (__clang_hip_math.h).  sqrtf() and '/' are IEEE under
// -fhip-fp32-correctly-rounded-divide-sqrt (on by default, set explicitly in the build).
__device__ __forceinline__ float ieee_sqrtf(float x) { return sqrtf(x); }
__device__ __forceinline__ float ieee_divf(float a, float b) { return a / b; }

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) {
  return __uint_as_float(((uint32_t)h) << 16);
}

__device__ __forceinline__ float f16_bits_to_f32(uint16_t h) {
  _Float16 x;
  __builtin_memcpy(&x, &h, 2);
  return (float)x;  // exact widening
}

__device__ __forceinline__ float load_elem(const void* base, uint32_t dtype, uint64_t i) {
  if (dtype == MI355_DTYPE_F32) return ((const float*)base)[i];
  if (dtype == MI355_DTYPE_BF16) return bf16_bits_to_f32(((const uint16_t*)base)[i]);
  return f16_bits_to_f32(((const uint16_t*)base)[i]);
}

// total order on (distance, rowid): python/python/lancedb/query.py:1368
__device__ __forceinline__ bool key_less(float d1, uint32_t hi1, uint32_t lo1, float d2,
                                         uint32_t hi2, uint32_t lo2) {
  if (d1 < d2) return true;
  if (d1 > d2) return false;
  if (hi1 != hi2) return hi1 < hi2;
  return lo1 < lo2;
}

// distance_range [lower, upper) (rust/lancedb/src/query.rs:1282-1288); NaN = NULL -> dropped
struct RangeFilter {
  uint32_t has_lower, has_upper;
  float lower, upper;
};

__device__ __forceinline__ bool in_range(float d, const RangeFilter& r) {
  if (d != d) return false;
  if (r.has_lower && !(d >= r.lower)) return false;
  if (r.has_upper && !(d < r.upper)) return false;
  return true;
}

// Device-decided batch size (maximum_nprobes second pass, query.rs:1246-1262): the queries that came
// back short are compacted on the device (k_compact_short) and the second pass is launched over the
// FULL slot count with this mask — slots at or past *n are inactive and cost an early exit, so the
// host never reads the count (no synchronisation inside a device-I/O call).
struct ActiveMask {
  const uint32_t* n = nullptr;  // device: number of active slots, or nullptr (every slot is active)
  uint32_t base = 0;            // slot index of this launch's query 0
  __device__ __forceinline__ bool on(uint32_t b) const { return !n || base + b < *n; }
};

// prefilter (mi355_search_params.filter_*): sorted unique row ids, allow or block list.
// Evaluated lazily, only for rows that already beat the running distance threshold.
struct RowFilter {
  uint32_t mode;  // MI355_FILTER_*
  uint32_t pad;
  const uint64_t* ids;
  uint64_t n;
};

__device__ __forceinline__ bool row_permitted(uint64_t id, const RowFilter& f) {
  if (f.mode == MI355_FILTER_NONE) return true;
  uint64_t lo = 0, hi = f.n;
  while (lo < hi) {
    const uint64_t mid = lo + ((hi - lo) >> 1);
    if (f.ids[mid] < id) lo = mid + 1; else hi = mid;
  }
  const bool found = lo < f.n && f.ids[lo] == id;
  return f.mode == MI355_FILTER_ALLOW ? found : !found;
}

__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane));
}
__device__ __forceinline__ uint32_t readlane_u(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}

// ---------------------------------------------------------------------------
// K4: wave-level top-kk reducer ("replace the worst").  The kk best keys seen
// so far live unsorted in registers, KPL per lane (slot g = s*64 + lane < kk);
// the current worst kept key is tracked wave-uniformly and is the admission
// threshold.  A candidate is admitted iff key < worst, so the kept set is
// always the exact kk smallest by (distance, rowid).  Expected admissions for n
// random keys are ~kk*ln(n/kk): the hot loop only pays one compare per row.
// ---------------------------------------------------------------------------
//
// k > 64 * KPL: the selector runs in PASSES (wave_select_sorted below).  Pass p admits
// only keys strictly above the `floor` = the last key pass p-1 emitted, so it yields ranks
// p*C .. p*C + C-1 of the same total order; the reference bounds neither `limit`
// (rust/lancedb/src/query.rs:818-907) nor `refine_factor` (query.rs:1302-1332).
template <int KPL>
struct WaveTopK {
  float d[KPL];
  uint32_t pos[KPL], lo[KPL], hi[KPL];
  float thr_d;  // worst kept key (wave-uniform)
  uint32_t thr_lo, thr_hi;
  int thr_lane, thr_slot;
  uint32_t kk;
  // admission floor (wave-uniform): keys <= floor are ignored
  uint32_t fl_on;
  float fl_d;
  uint32_t fl_lo, fl_hi;
  // last key handed out by drain_sorted (wave-uniform)
  float last_d;
  uint32_t last_lo, last_hi;

  __device__ __forceinline__ void set_floor(bool on, float fd, uint32_t flo, uint32_t fhi) {
    fl_on = on ? 1u : 0u;
    fl_d = fd;
    fl_lo = flo;
    fl_hi = fhi;
  }

  __device__ __forceinline__ void init(uint32_t kk_, int lane) {
    kk = kk_;
    fl_on = 0;
    fl_d = 0.f;
    fl_lo = fl_hi = 0;
    last_d = 0.f;
    last_lo = last_hi = 0;
#pragma unroll
    for (int s = 0; s < KPL; ++s) {
      d[s] = __builtin_huge_valf();
      pos[s] = CAND_EMPTY_POS;
      lo[s] = 0xFFFFFFFFu;
      hi[s] = 0xFFFFFFFFu;
    }
    recompute(lane);
  }

  // find the worst enabled slot across the wave (ties: lowest lane, lowest slot)
  __device__ __forceinline__ void recompute(int lane) {
    float bd = 0.f;
    uint32_t blo = 0, bhi = 0;
    int bs = 0, bv = 0, bl = lane;
#pragma unroll
    for (int s = 0; s < KPL; ++s) {
      bool en = (uint32_t)(s * MI355_WAVE + lane) < kk;
      if (en && (!bv || key_less(bd, bhi, blo, d[s], hi[s], lo[s]))) {
        bd = d[s];
        blo = lo[s];
        bhi = hi[s];
        bs = s;
        bv = 1;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      float od = __shfl_xor(bd, off);
      uint32_t olo = __shfl_xor(blo, off), ohi = __shfl_xor(bhi, off);
      int os = __shfl_xor(bs, off), ov = __shfl_xor(bv, off), ol = __shfl_xor(bl, off);
      bool take;
      if (!ov)
        take = false;
      else if (!bv)
        take = true;
      else if (key_less(bd, bhi, blo, od, ohi, olo))
        take = true;
      else if (key_less(od, ohi, olo, bd, bhi, blo))
        take = false;
      else
        take = (ol < bl) || (ol == bl && os < bs);
      if (take) {
        bd = od;
        blo = olo;
        bhi = ohi;
        bs = os;
        bv = ov;
        bl = ol;
      }
    }
    thr_d = bd;
    thr_lo = blo;
    thr_hi = bhi;
    thr_lane = bl;
    thr_slot = bs;
  }

  // all arguments wave-uniform
  __device__ __forceinline__ void insert_uniform(float cd, uint32_t cpos, uint32_t clo,
                                                 uint32_t chi, int lane) {
    if (!key_less(cd, chi, clo, thr_d, thr_hi, thr_lo)) return;
    if (fl_on && !key_less(fl_d, fl_hi, fl_lo, cd, chi, clo)) return;  // not above the floor
    if (lane == thr_lane) {
#pragma unroll
      for (int s = 0; s < KPL; ++s)
        if (s == thr_slot) {
          d[s] = cd;
          pos[s] = cpos;
          lo[s] = clo;
          hi[s] = chi;
        }
    }
    recompute(lane);
  }

  // Each lane offers (at most) one candidate; `valid` lanes whose distance can
  // still beat the threshold are drained one by one.
  __device__ __forceinline__ void offer(bool valid, float cd, uint32_t cpos, uint64_t cid,
                                        int lane) {
    uint64_t mask = __ballot(valid && cd <= thr_d && (!fl_on || cd >= fl_d));
    uint32_t clo = (uint32_t)cid, chi = (uint32_t)(cid >> 32);
    while (mask) {
      int l = __ffsll((unsigned long long)mask) - 1;
      mask &= mask - 1;
      insert_uniform(readlane_f(cd, l), readlane_u(cpos, l), readlane_u(clo, l),
                     readlane_u(chi, l), lane);
    }
  }

  // dump the kept keys (unsorted) to out[0..kk)
  __device__ __forceinline__ void store(Cand* out, int lane) const {
#pragma unroll
    for (int s = 0; s < KPL; ++s) {
      uint32_t g = (uint32_t)(s * MI355_WAVE + lane);
      if (g < kk) {
        Cand c;
        c.d = d[s];
        c.pos = pos[s];
        c.id = ((uint64_t)hi[s] << 32) | lo[s];
        out[g] = c;
      }
    }
  }

  // Emit the kept keys in ascending (distance, rowid) order: rank r goes to
  // emit(r, d, pos, id) on the owning lane.  Returns the number of real rows.
  template <typename F>
  __device__ __forceinline__ uint32_t drain_sorted(int lane, F emit) {
    uint32_t taken = 0;  // per-lane bitmask of emitted slots
    uint32_t n = 0;
    for (uint32_t r = 0; r < kk; ++r) {
      float bd = 0.f;
      uint32_t blo = 0, bhi = 0;
      int bs = 0, bv = 0, bl = lane;
#pragma unroll
      for (int s = 0; s < KPL; ++s) {
        bool en = (uint32_t)(s * MI355_WAVE + lane) < kk && !((taken >> s) & 1u) &&
                  pos[s] != CAND_EMPTY_POS;
        if (en && (!bv || key_less(d[s], hi[s], lo[s], bd, bhi, blo))) {
          bd = d[s];
          blo = lo[s];
          bhi = hi[s];
          bs = s;
          bv = 1;
        }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        float od = __shfl_xor(bd, off);
        uint32_t olo = __shfl_xor(blo, off), ohi = __shfl_xor(bhi, off);
        int os = __shfl_xor(bs, off), ov = __shfl_xor(bv, off), ol = __shfl_xor(bl, off);
        bool take;
        if (!ov)
          take = false;
        else if (!bv)
          take = true;
        else if (key_less(od, ohi, olo, bd, bhi, blo))
          take = true;
        else if (key_less(bd, bhi, blo, od, ohi, olo))
          take = false;
        else
          take = (ol < bl) || (ol == bl && os < bs);
        if (take) {
          bd = od;
          blo = olo;
          bhi = ohi;
          bs = os;
          bv = ov;
          bl = ol;
        }
      }
      if (!bv) break;  // wave-uniform: nothing left
      last_d = bd;  // after the butterfly every lane holds the winner's key
      last_lo = blo;
      last_hi = bhi;
      if (lane == bl) {
#pragma unroll
        for (int s = 0; s < KPL; ++s)
          if (s == bs) {
            emit(r, d[s], pos[s], ((uint64_t)hi[s] << 32) | lo[s]);
            taken |= 1u << s;
          }
      }
      ++n;
    }
    return n;
  }
};

// The k smallest keys of a candidate stream in ascending (distance, rowid) order, for ANY k:
// passes of up to 64 * KPL keys.  `gen(top)` must offer the WHOLE stream to `top` each time it
// is called (it runs once per pass); `emit(rank, d, pos, id)` runs on the owning lane.
// Returns the number of keys emitted (< k when the stream is shorter).  After the call
// `kth_d` holds the distance of the last emitted key (+inf when none).
template <int KPL, typename Gen, typename Emit>
__device__ __forceinline__ uint32_t wave_select_sorted(uint32_t k, int lane, Gen gen, Emit emit, float* kth_d = nullptr) {
  WaveTopK<KPL> top;
  uint32_t base = 0;
  bool fl_on = false;
  float fd = __builtin_huge_valf();
  uint32_t flo = 0, fhi = 0;
  while (base < k) {
    const uint32_t c = min(k - base, (uint32_t)(KPL * MI355_WAVE));
    top.init(c, lane);
    top.set_floor(fl_on, fd, flo, fhi);
    gen(top);
    const uint32_t b0 = base;
    const uint32_t n = top.drain_sorted(lane, [&](uint32_t r, float d, uint32_t pos, uint64_t id) { emit(b0 + r, d, pos, id); });
    base += n;
    if (n) {
      fl_on = true;
      fd = top.last_d;
      flo = top.last_lo;
      fhi = top.last_hi;
    }
    if (n < c) break;  // the stream is exhausted
  }
  if (kth_d) *kth_d = fl_on ? fd : __builtin_huge_valf();
  return base;
}

// ---------------------------------------------------------------------------
// K4 (scan-side): shuffle-free selection for the ADC hot loop.
//
//  * wave_kth_smallest_key: exact k-th smallest of the 64 lanes' keys by a
//    bitwise radix select on ballots (32 x {v_cmp, s_bcnt1}; no LDS, no DPP).
//    Applied to each lane's minimum over its rows it yields a threshold T' with
//    at least k rows <= T' in this wave, so the k best rows of the wave survive.
//  * WaveList: per-wave candidate list in LDS, 8 B per entry (distance, row
//    position).  Rows passing the threshold are appended with a ballot prefix;
//    when the list is full it is compacted to its exact kk best by rank
//    counting (every lane ranks its entries against a broadcast sweep of the
//    list).  Ties on the distance are broken by the row id, fetched lazily.
// Correct for any input (all-equal distances just compact more often);
// expected work per 1024 rows is ~kk appends and no compaction.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f32_sort_key(float d);

__device__ __forceinline__ float f32_from_sort_key(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(u);
}

// k is 1-based and must be <= 64; lanes that hold nothing pass key 0xFFFFFFFF
__device__ __forceinline__ uint32_t wave_kth_smallest_key(uint32_t key, uint32_t k) {
  uint64_t alive = ~0ull;
  uint32_t prefix = 0, need = k;
#pragma unroll
  for (int bit = 31; bit >= 0; --bit) {
    uint64_t zero = __ballot(((key >> bit) & 1u) == 0u) & alive;
    uint32_t c = (uint32_t)__popcll((unsigned long long)zero);
    if (c >= need) {
      alive = zero;
    } else {
      need -= c;
      alive &= ~zero;
      prefix |= 1u << bit;
    }
  }
  return prefix;
}

struct ListEnt {
  float d;
  uint32_t pos;
};

// QTRACK: every compaction also records the distance of the wave's q-th best row (t_q), the wave's
// share of a WORKGROUP-wide bound: if each of the NW waves holds q rows at or below its own t_q and
// NW * q >= kk, then kk rows lie at or below max_w t_q (k_scan_skew publishes it as the shared
// admission threshold: far tighter than any single wave's kk-th best when the winners are spread
// over the waves).
// FAST: compactions select instead of ranking and leave the kk best rows UNSORTED (compact_select); a compile-time property — as a
// run-time flag both forms were instantiated in every kernel and the scan's selection lambdas stopped being inlined
template <int R, bool QTRACK = false, bool FAST = false>  // capacity = 64 * R entries
struct WaveList {
  ListEnt* list;  // LDS
  uint32_t cnt;   // wave-uniform
  uint32_t kk;
  float t_run;    // distance of the kk-th best row seen so far (+inf until then)
  uint32_t q;     // (QTRACK) rank tracked by t_q
  float t_q;      // (QTRACK) distance of the q-th best row seen so far (+inf until q rows were compacted)

  static constexpr bool fast = FAST;  // callers of a FAST list use filter(), not prune()

  __device__ __forceinline__ void init(ListEnt* lds, uint32_t kk_, uint32_t q_ = 0) {
    list = lds;
    cnt = 0;
    kk = kk_;
    t_run = __builtin_huge_valf();
    q = q_;
    t_q = __builtin_huge_valf();
  }

  // The kk best entries by a radix select over the lanes' registers: 32 rounds of ballots find the kk-th smallest distance key T,
  // and the entries at or below it are packed to the front of the list in their old order — no LDS sweep.  (The rank-counting
  // compaction below reads the list entry by entry, one dependent LDS broadcast per entry: ~8 us for a full list of 128 rows.
  // A single query's work items all start without a bound, so every wave fills its list at once and compacted it twice per
  // item: 20 of a 43 us scan phase, round 6.)  Exact unless MORE rows than needed tie with the kk-th distance — then the row ids
  // decide and the caller falls back to the ranking form: returns false, nothing changed.
  __device__ __forceinline__ bool compact_select(int lane) {
    __threadfence_block();
    ListEnt mine[R];
    uint32_t key[R];
    bool in[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t slot = (uint32_t)(r * MI355_WAVE + lane);
      in[r] = slot < cnt;
      mine[r].d = 0.f;
      mine[r].pos = CAND_EMPTY_POS;
      if (in[r]) mine[r] = list[slot];
      key[r] = in[r] ? f32_sort_key(mine[r].d) : 0xFFFFFFFFu;
    }
    // the k-th smallest key of the list (k <= cnt) and how many of the entries EQUAL to it the k smallest include
    auto kth = [&](uint32_t k, uint32_t& need_eq) -> uint32_t {
      bool alive[R];
#pragma unroll
      for (int r = 0; r < R; ++r) alive[r] = in[r];
      uint32_t prefix = 0, need = k;
#pragma unroll
      for (int bit = 31; bit >= 0; --bit) {
        uint32_t c = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) c += (uint32_t)__popcll((unsigned long long)__ballot(alive[r] && ((key[r] >> bit) & 1u) == 0u));
        const bool take_zero = c >= need;
        if (!take_zero) {
          need -= c;
          prefix |= 1u << bit;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) alive[r] = alive[r] && ((((key[r] >> bit) & 1u) == 0u) == take_zero);
      }
      need_eq = need;
      return prefix;
    };
    if (cnt <= kk) {  // nothing to drop: only the q-th best is news (the list keeps its order)
      if (QTRACK && q > 1u && q <= cnt) {
        uint32_t ne;
        t_q = fminf(t_q, f32_from_sort_key(kth(q, ne)));
      }
      return true;
    }
    uint32_t need = 0;
    const uint32_t T = kth(kk, need);
    uint32_t n_eq = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) n_eq += (uint32_t)__popcll((unsigned long long)__ballot(in[r] && key[r] == T));
    if (n_eq != need) return false;  // a tie across the boundary: the row ids decide
    if (QTRACK && q > 1u && q <= kk) {  // (q = 1: the callers take the minimum of the list themselves — no second select)
      uint32_t ne;
      t_q = fminf(t_q, f32_from_sort_key(kth(q, ne)));  // (a distance: ties do not matter)
    }
    __threadfence_block();           // every entry is in registers before any slot is rewritten
    uint32_t base = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool k = in[r] && key[r] <= T;
      const uint64_t m = __ballot(k);
      if (k) list[base + (uint32_t)__popcll((unsigned long long)(m & ((1ull << lane) - 1ull)))] = mine[r];
      base += (uint32_t)__popcll((unsigned long long)m);
    }
    cnt = base;  // == kk
    t_run = fminf(t_run, f32_from_sort_key(T));
    __threadfence_block();
    return true;
  }

  // keep the kk best entries, sorted by (distance, id); idof(pos) -> row id  (`fast`: unsorted, see compact_select)
  template <typename IdOf>
  __device__ __forceinline__ void compact(int lane, IdOf idof) {
    if constexpr (FAST) {
      if (compact_select(lane)) return;
    }
    compact_rank(lane, idof);
  }
  template <typename IdOf>
  __device__ __forceinline__ void compact_rank(int lane, IdOf idof) {
    __threadfence_block();
    ListEnt mine[R];
    bool val[R];
    uint32_t rank[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      uint32_t slot = (uint32_t)(r * MI355_WAVE + lane);
      val[r] = slot < cnt;
      mine[r].d = __builtin_huge_valf();
      mine[r].pos = CAND_EMPTY_POS;
      if (val[r]) mine[r] = list[slot];
      rank[r] = 0;
    }
    for (uint32_t j = 0; j < cnt; ++j) {
      ListEnt c = list[j];  // same address in every lane: LDS broadcast
#pragma unroll
      for (int r = 0; r < R; ++r) {
        bool lt = c.d < mine[r].d;
        if (val[r] && c.d == mine[r].d && c.pos != mine[r].pos) lt = idof(c.pos) < idof(mine[r].pos);
        rank[r] += (val[r] && lt) ? 1u : 0u;
      }
    }
    __threadfence_block();
    float kth = __builtin_huge_valf(), qth = __builtin_huge_valf();
    bool has_kth = false, has_qth = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (val[r] && rank[r] < kk) list[rank[r]] = mine[r];
      if (val[r] && rank[r] == kk - 1) {
        has_kth = true;
        kth = mine[r].d;
      }
      if (QTRACK && val[r] && rank[r] == q - 1) {
        has_qth = true;
        qth = mine[r].d;
      }
    }
    uint64_t mk = __ballot(has_kth);
    if (mk) t_run = fminf(t_run, readlane_f(kth, __ffsll((unsigned long long)mk) - 1));
    if (QTRACK) {
      uint64_t mq = __ballot(has_qth);
      if (mq) t_q = fminf(t_q, readlane_f(qth, __ffsll((unsigned long long)mq) - 1));
    }
    cnt = min(cnt, kk);
    __threadfence_block();
  }

  // After compact() (the list is sorted by (distance, id)): drop every entry above `thr`.  The
  // kept entries are a prefix of the sorted list, so only the count changes.
  __device__ __forceinline__ void prune(float thr, int lane) {
    uint32_t keep = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t slot = (uint32_t)(r * MI355_WAVE + lane);
      const bool k = slot < cnt && list[slot].d <= thr;
      keep += (uint32_t)__popcll((unsigned long long)__ballot(k));
    }
    cnt = keep;
  }

  // Drop every entry above `thr` from a list in ANY order (the survivors keep their relative order).
  __device__ __forceinline__ void filter(float thr, int lane) {
    ListEnt mine[R];
    bool k[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t slot = (uint32_t)(r * MI355_WAVE + lane);
      k[r] = slot < cnt;
      mine[r].d = 0.f;
      mine[r].pos = CAND_EMPTY_POS;
      if (k[r]) mine[r] = list[slot];
      k[r] = k[r] && mine[r].d <= thr;
    }
    __threadfence_block();  // every entry is in registers before any slot is rewritten
    uint32_t base = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint64_t m = __ballot(k[r]);
      if (k[r]) list[base + (uint32_t)__popcll((unsigned long long)(m & ((1ull << lane) - 1ull)))] = mine[r];
      base += (uint32_t)__popcll((unsigned long long)m);
    }
    cnt = base;
    __threadfence_block();
  }

  // append() for a caller that made room itself (cnt <= capacity - 64 on entry): no compaction inside
  __device__ __forceinline__ void append_room(bool ok, float d, uint32_t pos, int lane) {
    const uint64_t mask = __ballot(ok);
    if (!mask) return;
    if (ok) {
      const uint32_t idx = cnt + (uint32_t)__popcll((unsigned long long)(mask & ((1ull << lane) - 1ull)));
      ListEnt e;
      e.d = d;
      e.pos = pos;
      list[idx] = e;
    }
    cnt += (uint32_t)__popcll((unsigned long long)mask);
  }

  // Every lane offers at most one row.  `thr` is the caller's current
  // admission threshold (updated when a compaction tightens it).
  template <typename IdOf>
  __device__ __forceinline__ void append(bool ok, float d, uint32_t pos, float& thr, int lane,
                                         IdOf idof) {
    uint64_t mask = __ballot(ok);
    if (!mask) return;
    uint32_t n = (uint32_t)__popcll((unsigned long long)mask);
    if (cnt + n > (uint32_t)(R * MI355_WAVE)) {
      compact(lane, idof);
      thr = fminf(thr, t_run);
      ok = ok && d <= thr;
      mask = __ballot(ok);
      n = (uint32_t)__popcll((unsigned long long)mask);
    }
    if (ok) {
      uint32_t idx = cnt + (uint32_t)__popcll((unsigned long long)(mask & ((1ull << lane) - 1ull)));
      ListEnt e;
      e.d = d;
      e.pos = pos;
      list[idx] = e;
    }
    cnt += n;
  }
};

// order-preserving u32 key of an f32 (NaN last; -0 == +0)
__device__ __forceinline__ uint32_t f32_sort_key(float d) {
  if (d != d) return 0xFFFFFFFFu;
  d = d + 0.0f;  // -0 -> +0 (not foldable under IEEE)
  uint32_t u = __float_as_uint(d);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
