// mi355_ann.hip — C-ABI implementation (include/mi355_ann.h) of the MI355X ANN
// scan engine: host-side planner + launches of the gfx950 kernels.
//
// Replaces, behind lancedb::query::VectorQuery, what
// /root/reference/rust/lancedb/src/table/query.rs:219-327 hands to the lance
// Scanner (E1 in SURVEY.md §2b): nearest / nprobes / refine / distance_range /
// use_index -> a fixed launch sequence per query batch instead of a DataFusion
// plan.  No torch, no Triton, no CUDA-compat headers; HIP runtime only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/mi355_ann.h"
#include "kernels_flat.h"
#include "kernels_flat_mfma.h"
#include "kernels_flat_mfma8.h"
#include "kernels_ivfpq.h"
#include "kernels_skew.h"
#include "kernels_encode.h"

// ------------------------------------------------------------------ errors --
static thread_local std::string g_last_error;

static int32_t fail(int32_t code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess)                                                               \
      return fail(MI355_ERR_RUNTIME, "HIP error %d (%s) at %s:%d: %s", (int)_e,         \
                  hipGetErrorString(_e), __FILE__, __LINE__, #expr);                    \
  } while (0)

#define ST_TRY(expr)            \
  do {                          \
    int32_t _s = (expr);        \
    if (_s != MI355_OK) return _s; \
  } while (0)

// grow-only device buffer
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int32_t ensure(size_t bytes) {
    if (bytes <= cap) return MI355_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      p = nullptr;
      return fail(MI355_ERR_RUNTIME, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    cap = want;
    return MI355_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T* as() const {
    return (T*)p;
  }
};

static uint32_t env_u32(const char* name, uint32_t dflt) {
  const char* s = getenv(name);
  if (!s || !*s) return dflt;
  return (uint32_t)strtoul(s, nullptr, 10);
}

// ---------------------------------------------------------------- handles ---
// Per-launch-sequence timestamps, recorded on the search stream without any
// host synchronisation; elapsed times are read back in mi355_last_stats.
struct EventSet {
  hipEvent_t ev[6];
};

struct mi355_index {
  int32_t device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::mutex mu;
  // shape
  uint32_t dim = 0, nlist = 0, m = 0, dsub = 0, metric = 0;
  uint64_t n_local = 0;
  uint32_t parts_owned = 0, max_len = 0;
  uint32_t shard_count = 1, shard_rank = 0;
  // device data
  DevBuf centroids, cnorm, codebook, codes, code_off, plen, pstride, lrow0, grow0, row_ids, raw;
  bool has_row_ids = false, has_raw = false;
  uint32_t raw_dtype = 0;
  std::vector<uint32_t> h_plen;
  // code layout: MI355_SCAN_PAIR = [m][pstride] blocks, MI355_SCAN_SKEW = pre-skewed streams
  uint32_t layout = MI355_SCAN_PAIR;
  uint32_t n_cus = 256;
  DevBuf cbT, order, xcd_first, p_cnt, p_off, p_fill, q_start, heads, items, qthr, w_filter, w_bad, w_probes64;
  // workspace
  DevBuf w_q, w_qp, w_qq, w_coarse, w_probes, w_cand, w_ids, w_dist, w_pos, w_cnt, w_ids2,
      w_dist2, w_cnt2, w_stat;
  // config
  uint32_t scan_variant = MI355_SCAN_AUTO, slice_rows = 0, profile = 0;
  mi355_stats stats{};
  std::vector<EventSet> ev_free, ev_pending;
};

struct mi355_flat {
  int32_t device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::mutex mu;
  uint32_t dim = 0, dtype = 0;
  uint64_t n_rows = 0;
  DevBuf vectors, row_ids;
  const void* borrowed_vectors = nullptr;
  const uint64_t* borrowed_ids = nullptr;
  bool has_row_ids = false;
  DevBuf w_q, w_cand, w_ids, w_dist, w_cnt;
  // MFMA filter + exact re-rank (kernels_flat_mfma.h)
  bool mfma = false;       // built at open when the column is large enough
  bool shadowed = false;   // GEMM reads a bf16 shadow (column is f32/f16 or dim % 64 != 0)
  uint32_t dimp = 0;
  float c_err = 0.f, vv_max = 0.f;
  DevBuf shadow, vv, vmax, g_qb, g_qa, g_qg, g_slack, g_tau, g_gm, g_seg, g_cnt, g_cand, w_filter;
  uint32_t last_path = 0;  // 1 = MFMA filter, 2 = exact sweep (reported by mi355_flat_info)
};

static int32_t drain_events(mi355_index* ix, bool discard);
static void reset_stats(mi355_index* ix);

static IndexView make_view(const mi355_index* ix) {
  IndexView v;
  v.dim = ix->dim;
  v.nlist = ix->nlist;
  v.m = ix->m;
  v.dsub = ix->dsub;
  v.metric = ix->metric;
  v.centroids = ix->centroids.as<float>();
  v.cnorm = ix->cnorm.as<float>();
  v.codebook = ix->codebook.as<float>();
  v.codes = ix->codes.as<uint8_t>();
  v.code_off = ix->code_off.as<uint64_t>();
  v.plen = ix->plen.as<uint32_t>();
  v.pstride = ix->pstride.as<uint32_t>();
  v.lrow0 = ix->lrow0.as<uint32_t>();
  v.grow0 = ix->grow0.as<uint64_t>();
  v.row_ids = ix->has_row_ids ? ix->row_ids.as<uint64_t>() : nullptr;
  v.raw = ix->has_raw ? ix->raw.p : nullptr;
  v.raw_dtype = ix->raw_dtype;
  return v;
}

static size_t dtype_size(uint32_t dt) { return dt == MI355_DTYPE_F32 ? 4 : 2; }

// copy `bytes` from a caller buffer (host or device) to device memory
static hipError_t copy_in(void* dst, const void* src, size_t bytes, uint32_t mem, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  return hipMemcpyAsync(dst, src, bytes,
                        mem == MI355_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                        s);
}

// ------------------------------------------------------------ shard plan ----
static void shard_plan_host(const uint64_t* po, uint32_t nlist, uint32_t shards,
                            std::vector<uint32_t>& owner) {
  owner.assign(nlist, 0);
  if (shards <= 1) return;
  std::vector<uint32_t> order(nlist);
  for (uint32_t p = 0; p < nlist; ++p) order[p] = p;
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    uint64_t la = po[a + 1] - po[a], lb = po[b + 1] - po[b];
    if (la != lb) return la > lb;
    return a < b;
  });
  std::vector<uint64_t> load(shards, 0);
  for (uint32_t i = 0; i < nlist; ++i) {
    uint32_t p = order[i], best = 0;
    for (uint32_t s = 1; s < shards; ++s)
      if (load[s] < load[best]) best = s;
    owner[p] = best;
    load[best] += po[p + 1] - po[p];
  }
}

// ---------------------------------------------------------------- library ---
extern "C" uint32_t mi355_abi_version(void) { return MI355_ANN_ABI_VERSION; }

extern "C" int32_t mi355_device_count(int32_t* out_count) {
  if (!out_count) return fail(MI355_ERR_INVALID_INPUT, "out_count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *out_count = n;
  return MI355_OK;
}

extern "C" int32_t mi355_last_error(char* buf, size_t buf_len) {
  if (!buf || buf_len == 0) return MI355_ERR_INVALID_INPUT;
  snprintf(buf, buf_len, "%s", g_last_error.c_str());
  return MI355_OK;
}

extern "C" int32_t mi355_shard_plan(const uint64_t* part_offsets, uint32_t nlist,
                                    uint32_t shard_count, uint32_t* out_owner) {
  if (!part_offsets || !out_owner || shard_count == 0 || nlist == 0)
    return fail(MI355_ERR_INVALID_INPUT, "mi355_shard_plan: bad arguments");
  std::vector<uint32_t> owner;
  shard_plan_host(part_offsets, nlist, shard_count, owner);
  memcpy(out_owner, owner.data(), sizeof(uint32_t) * nlist);
  return MI355_OK;
}

static int32_t need_device(int32_t device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(MI355_ERR_RUNTIME,
                "no HIP device visible (hipGetDeviceCount: %s); the MI355X engine has no CPU "
                "fallback",
                e == hipSuccess ? "0 devices" : hipGetErrorString(e));
  }
  if (device < 0 || device >= n)
    return fail(MI355_ERR_INVALID_INPUT, "device %d out of range (0..%d)", device, n - 1);
  HIP_TRY(hipSetDevice(device));
  return MI355_OK;
}

// ------------------------------------------------------------- index open ---
static int32_t validate_index_desc(const mi355_index_desc* d) {
  if (!d) return fail(MI355_ERR_INVALID_INPUT, "desc is NULL");
  if (d->struct_size != sizeof(mi355_index_desc))
    return fail(MI355_ERR_INVALID_INPUT, "mi355_index_desc.struct_size %u != %zu (ABI mismatch)",
                d->struct_size, sizeof(mi355_index_desc));
  if (d->nbits == 4)
    return fail(MI355_ERR_NOT_SUPPORTED, "4-bit PQ is not supported yet (only num_bits=8)");
  if (d->nbits != 8) return fail(MI355_ERR_INVALID_INPUT, "num_bits must be 4 or 8, got %u", d->nbits);
  if (d->dim == 0 || d->nlist == 0 || d->m == 0)
    return fail(MI355_ERR_INVALID_INPUT, "dim, nlist and m must be > 0");
  if (d->dim % d->m != 0)
    return fail(MI355_ERR_INVALID_INPUT, "dim %u is not divisible by num_sub_vectors %u", d->dim, d->m);
  if (d->metric > MI355_METRIC_DOT)
    return fail(MI355_ERR_INVALID_INPUT, "unknown metric %u", d->metric);
  if (d->mem > MI355_MEM_DEVICE || d->codes_layout > MI355_CODES_PART_TRANSPOSED ||
      d->raw_dtype > MI355_DTYPE_F16)
    return fail(MI355_ERR_INVALID_INPUT, "bad mem / codes_layout / raw_dtype enum");
  if (!d->centroids || !d->codebook || !d->part_offsets)
    return fail(MI355_ERR_INVALID_INPUT, "centroids, codebook and part_offsets are required");
  if (d->n_rows && !d->codes) return fail(MI355_ERR_INVALID_INPUT, "codes is NULL");
  if (d->part_offsets[0] != 0 || d->part_offsets[d->nlist] != d->n_rows)
    return fail(MI355_ERR_INVALID_INPUT, "part_offsets must run from 0 to n_rows");
  for (uint32_t p = 0; p < d->nlist; ++p) {
    if (d->part_offsets[p + 1] < d->part_offsets[p])
      return fail(MI355_ERR_INVALID_INPUT, "part_offsets must be non-decreasing");
    if (d->part_offsets[p + 1] - d->part_offsets[p] >= 0xFFFFFFF0ull)
      return fail(MI355_ERR_NOT_SUPPORTED, "partition %u has >= 2^32 rows", p);
  }
  if (d->shard_count > 1 && d->shard_rank >= d->shard_count)
    return fail(MI355_ERR_INVALID_INPUT, "shard_rank %u >= shard_count %u", d->shard_rank,
                d->shard_count);
  if ((size_t)d->m * 1024 + (size_t)d->dim * 4 > 160u * 1024)
    return fail(MI355_ERR_NOT_SUPPORTED,
                "distance table of %u sub-vectors (%u KiB) does not fit the 160 KiB LDS", d->m, d->m);
  return MI355_OK;
}

static int32_t index_free(mi355_index* ix) {
  if (!ix) return MI355_OK;
  (void)hipSetDevice(ix->device);
  DevBuf* bufs[] = {&ix->centroids, &ix->cnorm,  &ix->codebook, &ix->codes,   &ix->code_off,
                    &ix->plen,      &ix->pstride, &ix->lrow0,    &ix->grow0,   &ix->row_ids,
                    &ix->raw,       &ix->w_q,    &ix->w_qp,     &ix->w_qq,    &ix->w_coarse,
                    &ix->w_probes,  &ix->w_cand, &ix->w_ids,    &ix->w_dist,  &ix->w_pos,
                    &ix->w_cnt,     &ix->w_ids2, &ix->w_dist2,  &ix->w_cnt2,  &ix->w_stat,
                    &ix->cbT,       &ix->order,  &ix->xcd_first, &ix->p_cnt,  &ix->p_off,
                    &ix->p_fill,    &ix->q_start, &ix->heads,   &ix->items,   &ix->qthr,
                    &ix->w_filter,  &ix->w_bad,    &ix->w_probes64};
  for (DevBuf* b : bufs) b->release();
  for (auto* v : {&ix->ev_free, &ix->ev_pending})
    for (auto& es : *v)
      for (auto& e : es.ev) (void)hipEventDestroy(e);
  if (ix->own_stream) (void)hipStreamDestroy(ix->own_stream);
  delete ix;
  return MI355_OK;
}

static int32_t index_open_impl(const mi355_index_desc* d, mi355_index* ix) {
  ix->device = d->device;
  ix->dim = d->dim;
  ix->nlist = d->nlist;
  ix->m = d->m;
  ix->dsub = d->dim / d->m;
  ix->metric = d->metric;
  ix->shard_count = d->shard_count > 1 ? d->shard_count : 1;
  ix->shard_rank = d->shard_count > 1 ? d->shard_rank : 0;
  HIP_TRY(hipStreamCreateWithFlags(&ix->own_stream, hipStreamNonBlocking));
  ix->stream = ix->own_stream;
  hipStream_t st = ix->stream;
  const uint32_t nlist = d->nlist, m = d->m;

  // -- ownership + local layout
  std::vector<uint32_t> owner;
  shard_plan_host(d->part_offsets, nlist, ix->shard_count, owner);
  std::vector<uint32_t> plen(nlist), pstride(nlist), lrow0(nlist);
  std::vector<uint64_t> code_off(nlist), grow0(nlist);
  uint64_t rows = 0, bytes = 0;
  uint32_t owned = 0, max_len = 0;
  {
    const char* lay = getenv("MI355_LAYOUT");  // dev knob: "pair" forces the generic layout
    const bool force_pair = lay && !strcmp(lay, "pair");
    // the skewed layout needs the 256 x P-dword table + residual + lists in 160 KiB of LDS
    const size_t lds_skew = (size_t)SK_TABLE_BYTES + (size_t)d->dim * 4 + 25 * 1024;
    ix->layout = (!force_pair && sk_supported_m(m) && d->dim <= 2048 && lds_skew <= 160u * 1024) ? MI355_SCAN_SKEW : MI355_SCAN_PAIR;
  }
  const bool skew = ix->layout == MI355_SCAN_SKEW;
  for (uint32_t p = 0; p < nlist; ++p) {
    uint64_t len = d->part_offsets[p + 1] - d->part_offsets[p];
    bool mine = owner[p] == ix->shard_rank;
    plen[p] = mine ? (uint32_t)len : 0;
    pstride[p] = (plen[p] + 15u) & ~15u;
    lrow0[p] = (uint32_t)rows;
    grow0[p] = d->part_offsets[p];
    code_off[p] = bytes;
    rows += plen[p];
    bytes += skew ? sk_part_chunks((plen[p] + SK_TILE - 1) / SK_TILE, m / 16) * 1024u : (uint64_t)m * pstride[p];
    if (plen[p]) {
      ++owned;
      max_len = std::max(max_len, plen[p]);
    }
  }
  if (rows >= 0xFFFFFFF0ull)
    return fail(MI355_ERR_NOT_SUPPORTED, "%llu rows on one handle (limit 2^32-16); shard the index",
                (unsigned long long)rows);
  ix->n_local = rows;
  ix->parts_owned = owned;
  ix->max_len = max_len;
  ix->h_plen = plen;

  // -- small tables
  ST_TRY(ix->centroids.ensure(sizeof(float) * (size_t)nlist * d->dim));
  ST_TRY(ix->cnorm.ensure(sizeof(float) * nlist));
  ST_TRY(ix->codebook.ensure(sizeof(float) * (size_t)m * 256 * ix->dsub));
  ST_TRY(ix->code_off.ensure(sizeof(uint64_t) * nlist));
  ST_TRY(ix->plen.ensure(sizeof(uint32_t) * nlist));
  ST_TRY(ix->pstride.ensure(sizeof(uint32_t) * nlist));
  ST_TRY(ix->lrow0.ensure(sizeof(uint32_t) * nlist));
  ST_TRY(ix->grow0.ensure(sizeof(uint64_t) * nlist));
  HIP_TRY(copy_in(ix->centroids.p, d->centroids, sizeof(float) * (size_t)nlist * d->dim, d->mem, st));
  HIP_TRY(copy_in(ix->codebook.p, d->codebook, sizeof(float) * (size_t)m * 256 * ix->dsub, d->mem, st));
  HIP_TRY(hipMemcpyAsync(ix->code_off.p, code_off.data(), sizeof(uint64_t) * nlist, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(ix->plen.p, plen.data(), sizeof(uint32_t) * nlist, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(ix->pstride.p, pstride.data(), sizeof(uint32_t) * nlist, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(ix->lrow0.p, lrow0.data(), sizeof(uint32_t) * nlist, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(ix->grow0.p, grow0.data(), sizeof(uint64_t) * nlist, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_centroid_norms, dim3((nlist + 63) / 64), dim3(64), 0, st,
                     ix->centroids.as<float>(), nlist, d->dim, ix->cnorm.as<float>());
  HIP_TRY(hipGetLastError());

  // -- PQ codes: stage (host source) and re-pack into [m][pstride] blocks
  ST_TRY(ix->codes.ensure(bytes + 64));
  if (rows) {
    const size_t STAGE = (size_t)env_u32("MI355_STAGE_MB", 256) << 20;
    DevBuf stage, d_srcoff, d_pids;
    std::vector<uint64_t> srcoff;
    std::vector<uint32_t> pids;
    auto flush = [&](uint32_t batch_max_stride) -> int32_t {
      if (pids.empty()) return MI355_OK;
      ST_TRY(d_srcoff.ensure(sizeof(uint64_t) * pids.size()));
      ST_TRY(d_pids.ensure(sizeof(uint32_t) * pids.size()));
      HIP_TRY(hipMemcpyAsync(d_srcoff.p, srcoff.data(), sizeof(uint64_t) * pids.size(), hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemcpyAsync(d_pids.p, pids.data(), sizeof(uint32_t) * pids.size(), hipMemcpyHostToDevice, st));
      RepackArgs ra;
      ra.src = d->mem == MI355_MEM_DEVICE ? d->codes : stage.as<uint8_t>();
      ra.src_off = d_srcoff.as<uint64_t>();
      ra.part_ids = d_pids.as<uint32_t>();
      ra.dst = ix->codes.as<uint8_t>();
      ra.code_off = ix->code_off.as<uint64_t>();
      ra.plen = ix->plen.as<uint32_t>();
      ra.pstride = ix->pstride.as<uint32_t>();
      ra.m = m;
      ra.transposed = d->codes_layout == MI355_CODES_PART_TRANSPOSED;
      SkewPackArgs sp;
      sp.src = ra.src;
      sp.src_off = ra.src_off;
      sp.part_ids = ra.part_ids;
      sp.dst = ra.dst;
      sp.code_off = ra.code_off;
      sp.plen = ra.plen;
      sp.m = m;
      sp.transposed = ra.transposed;
      // grid.y is limited to 65535: split very wide batches
      for (size_t y0 = 0; y0 < pids.size(); y0 += 32768) {
        uint32_t ny = (uint32_t)std::min<size_t>(32768, pids.size() - y0);
        if (skew) {
          SkewPackArgs sb = sp;
          sb.src_off += y0;
          sb.part_ids += y0;
          hipLaunchKernelGGL(k_pack_skew, dim3(sk_pack_slots(batch_max_stride), ny), dim3(256),
                             2 * 64 * (m + 1), st, sb);
        } else {
          RepackArgs rb = ra;
          rb.src_off += y0;
          rb.part_ids += y0;
          hipLaunchKernelGGL(k_repack_codes, dim3((batch_max_stride + 63) / 64, ny), dim3(256),
                             64 * (m + 1), st, rb);
        }
        HIP_TRY(hipGetLastError());
      }
      HIP_TRY(hipStreamSynchronize(st));  // staging buffer / host vectors are reused
      srcoff.clear();
      pids.clear();
      return MI355_OK;
    };
    if (d->mem == MI355_MEM_HOST) ST_TRY(stage.ensure(STAGE));
    size_t used = 0;
    uint32_t bmax = 0;
    for (uint32_t p = 0; p < nlist; ++p) {
      if (!plen[p]) continue;
      size_t pbytes = (size_t)m * plen[p];
      uint64_t soff = (uint64_t)m * d->part_offsets[p];
      if (d->mem == MI355_MEM_HOST) {
        if (pbytes > STAGE) {  // a partition larger than the staging buffer: grow once
          ST_TRY(flush(bmax));
          used = 0;
          bmax = 0;
          ST_TRY(stage.ensure(pbytes));
        }
        if (used + pbytes > stage.cap) {
          ST_TRY(flush(bmax));
          used = 0;
          bmax = 0;
        }
        HIP_TRY(hipMemcpyAsync(stage.as<uint8_t>() + used, d->codes + soff, pbytes, hipMemcpyHostToDevice, st));
        srcoff.push_back(used);
        used += (pbytes + 15) & ~(size_t)15;
      } else {
        srcoff.push_back(soff);
      }
      pids.push_back(p);
      bmax = std::max(bmax, pstride[p]);
    }
    ST_TRY(flush(bmax));
    stage.release();
    d_srcoff.release();
    d_pids.release();
  }

  // -- skewed layout: transposed codebook, static partition order and planner buffers
  if (skew) {
    const size_t cb_elems = (size_t)m * 256 * ix->dsub;
    ST_TRY(ix->cbT.ensure(sizeof(float) * cb_elems));
    hipLaunchKernelGGL(k_transpose_codebook, dim3((uint32_t)((cb_elems + 255) / 256)), dim3(256), 0, st,
                       ix->codebook.as<float>(), m, ix->dsub, ix->cbT.as<float>());
    HIP_TRY(hipGetLastError());
    // Queue x (the XCD that scans it first) gets partitions by greedy
    // longest-first bin packing; inside a queue the longest partitions go first
    // so that the tail of a batch is made of short work items.
    std::vector<uint32_t> by_len(nlist);
    for (uint32_t p = 0; p < nlist; ++p) by_len[p] = p;
    std::sort(by_len.begin(), by_len.end(), [&](uint32_t a, uint32_t b) {
      if (plen[a] != plen[b]) return plen[a] > plen[b];
      return a < b;
    });
    std::vector<std::vector<uint32_t>> queue(8);
    uint64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t i = 0; i < nlist; ++i) {
      uint32_t p = by_len[i], best = 0;
      for (uint32_t x = 1; x < 8; ++x)
        if (load[x] < load[best]) best = x;
      queue[best].push_back(p);
      load[best] += plen[p] + 1;  // +1: spread empty partitions too
    }
    std::vector<uint32_t> order, xcd_first(9);
    for (uint32_t x = 0; x < 8; ++x) {
      xcd_first[x] = (uint32_t)order.size();
      order.insert(order.end(), queue[x].begin(), queue[x].end());
    }
    xcd_first[8] = nlist;
    ST_TRY(ix->order.ensure(sizeof(uint32_t) * nlist));
    ST_TRY(ix->xcd_first.ensure(sizeof(uint32_t) * 9));
    ST_TRY(ix->p_cnt.ensure(sizeof(uint32_t) * nlist));
    ST_TRY(ix->p_off.ensure(sizeof(uint32_t) * nlist));
    ST_TRY(ix->p_fill.ensure(sizeof(uint32_t) * nlist));
    ST_TRY(ix->q_start.ensure(sizeof(uint32_t) * 16));
    ST_TRY(ix->heads.ensure(sizeof(uint32_t) * 8 * SK_HEAD_STRIDE));
    HIP_TRY(hipMemcpyAsync(ix->order.p, order.data(), sizeof(uint32_t) * nlist, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ix->xcd_first.p, xcd_first.data(), sizeof(uint32_t) * 9, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(ix->p_cnt.p, 0, sizeof(uint32_t) * nlist, st));
    HIP_TRY(hipStreamSynchronize(st));  // host vectors above go out of scope
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ix->device));
    ix->n_cus = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
  }

  // -- row ids and raw vectors: owned partitions, concatenated in local order
  auto gather_rows = [&](DevBuf& dst, const void* src, size_t row_bytes) -> int32_t {
    ST_TRY(dst.ensure(std::max<size_t>(row_bytes * rows, 16)));
    uint32_t p = 0;
    while (p < nlist) {
      if (!plen[p]) {
        ++p;
        continue;
      }
      uint32_t e = p;  // extend over a run of consecutive owned partitions
      uint64_t run = 0;
      while (e < nlist && (plen[e] || d->part_offsets[e + 1] == d->part_offsets[e])) {
        run += plen[e];
        ++e;
      }
      HIP_TRY(copy_in((uint8_t*)dst.p + (size_t)lrow0[p] * row_bytes,
                      (const uint8_t*)src + (size_t)d->part_offsets[p] * row_bytes,
                      (size_t)run * row_bytes, d->mem, st));
      p = e;
    }
    return MI355_OK;
  };
  if (d->row_ids) {
    ST_TRY(gather_rows(ix->row_ids, d->row_ids, sizeof(uint64_t)));
    ix->has_row_ids = true;
  }
  if (d->raw_vectors) {
    ST_TRY(gather_rows(ix->raw, d->raw_vectors, dtype_size(d->raw_dtype) * d->dim));
    ix->has_raw = true;
    ix->raw_dtype = d->raw_dtype;
  }
  ST_TRY(ix->w_stat.ensure(64));
  ST_TRY(ix->w_bad.ensure(64));
  HIP_TRY(hipStreamSynchronize(st));
  return MI355_OK;
}

extern "C" int32_t mi355_index_open(const mi355_index_desc* desc, mi355_index** out) {
  if (!out) return fail(MI355_ERR_INVALID_INPUT, "out is NULL");
  *out = nullptr;
  ST_TRY(validate_index_desc(desc));
  ST_TRY(need_device(desc->device));
  mi355_index* ix = new (std::nothrow) mi355_index();
  if (!ix) return fail(MI355_ERR_RUNTIME, "out of host memory");
  int32_t s = index_open_impl(desc, ix);
  if (s != MI355_OK) {
    index_free(ix);
    return s;
  }
  *out = ix;
  return MI355_OK;
}

extern "C" int32_t mi355_index_close(mi355_index* index) { return index_free(index); }

extern "C" int32_t mi355_index_set_stream(mi355_index* ix, void* hip_stream) {
  if (!ix) return fail(MI355_ERR_INVALID_INPUT, "index is NULL");
  std::lock_guard<std::mutex> lk(ix->mu);
  ix->stream = hip_stream ? (hipStream_t)hip_stream : ix->own_stream;
  return MI355_OK;
}

extern "C" int32_t mi355_index_sync(mi355_index* ix) {
  if (!ix) return fail(MI355_ERR_INVALID_INPUT, "index is NULL");
  HIP_TRY(hipSetDevice(ix->device));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  return MI355_OK;
}

extern "C" int32_t mi355_index_configure(mi355_index* ix, uint32_t scan_variant,
                                         uint32_t slice_rows, uint32_t profile) {
  if (!ix) return fail(MI355_ERR_INVALID_INPUT, "index is NULL");
  if (scan_variant > MI355_SCAN_SKEW) return fail(MI355_ERR_INVALID_INPUT, "unknown scan variant");
  if (scan_variant != MI355_SCAN_AUTO && scan_variant != ix->layout)
    return fail(MI355_ERR_INVALID_INPUT,
                "scan variant %u does not match the code layout this index was packed for (%u)",
                scan_variant, ix->layout);
  std::lock_guard<std::mutex> lk(ix->mu);
  ix->scan_variant = scan_variant;
  ix->slice_rows = (slice_rows + 15u) & ~15u;
  ix->profile = profile;
  HIP_TRY(hipSetDevice(ix->device));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  ST_TRY(drain_events(ix, true));
  reset_stats(ix);
  HIP_TRY(hipMemset(ix->w_stat.p, 0, 64));
  return MI355_OK;
}

extern "C" int32_t mi355_index_info(const mi355_index* ix, uint64_t* out_rows,
                                    uint32_t* out_partitions_owned) {
  if (!ix) return fail(MI355_ERR_INVALID_INPUT, "index is NULL");
  if (out_rows) *out_rows = ix->n_local;
  if (out_partitions_owned) *out_partitions_owned = ix->parts_owned;
  return MI355_OK;
}

// fold the pending timestamps into the stats (waits for the recorded work)
static int32_t drain_events(mi355_index* ix, bool discard) {
  for (auto& es : ix->ev_pending) {
    HIP_TRY(hipEventSynchronize(es.ev[5]));
    if (!discard) {
      float us[5];
      for (int i = 0; i < 5; ++i) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, es.ev[i], es.ev[i + 1]));
        us[i] = ms * 1000.f;
      }
      ix->stats.us_coarse += us[0];
      ix->stats.us_select += us[1];
      ix->stats.us_scan += us[2];
      ix->stats.us_merge += us[3];
      ix->stats.us_refine += us[4];
      ix->stats.us_total += us[0] + us[1] + us[2] + us[3] + us[4];
      ix->stats.scan_launches += 1;
    }
    ix->ev_free.push_back(es);
  }
  ix->ev_pending.clear();
  return MI355_OK;
}

static void reset_stats(mi355_index* ix) {
  ix->stats = mi355_stats{};
  ix->stats.struct_size = sizeof(mi355_stats);
}

extern "C" int32_t mi355_last_stats(mi355_index* ix, mi355_stats* out) {
  if (!ix || !out) return fail(MI355_ERR_INVALID_INPUT, "NULL argument");
  if (out->struct_size != sizeof(mi355_stats))
    return fail(MI355_ERR_INVALID_INPUT, "mi355_stats.struct_size mismatch");
  std::lock_guard<std::mutex> lk(ix->mu);
  HIP_TRY(hipSetDevice(ix->device));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  ST_TRY(drain_events(ix, false));
  unsigned long long rows = 0;
  HIP_TRY(hipMemcpy(&rows, ix->w_stat.p, sizeof(rows), hipMemcpyDeviceToHost));
  ix->stats.vectors_scanned = rows;
  ix->stats.code_bytes_scanned = rows * ix->m;
  *out = ix->stats;
  out->struct_size = sizeof(mi355_stats);
  return MI355_OK;
}

// ------------------------------------------------------------------ search --
static int32_t validate_params(const mi355_search_params* p) {
  if (!p) return fail(MI355_ERR_INVALID_INPUT, "params is NULL");
  if (p->struct_size != sizeof(mi355_search_params))
    return fail(MI355_ERR_INVALID_INPUT, "mi355_search_params.struct_size %u != %zu (ABI mismatch)",
                p->struct_size, sizeof(mi355_search_params));
  if (p->io_mem > MI355_MEM_DEVICE) return fail(MI355_ERR_INVALID_INPUT, "bad io_mem");
  if (p->filter_mode > MI355_FILTER_BLOCK) return fail(MI355_ERR_INVALID_INPUT, "unknown filter_mode %u", p->filter_mode);
  if (p->filter_mode != MI355_FILTER_NONE && p->n_filter && !p->filter_rowids)
    return fail(MI355_ERR_INVALID_INPUT, "filter_rowids is NULL");
  return MI355_OK;
}

// device view of the prefilter; a host array is staged into `stage`
static int32_t make_row_filter(const mi355_search_params* p, DevBuf& stage, hipStream_t st, RowFilter* out) {
  out->mode = p->filter_mode;
  out->pad = 0;
  out->ids = nullptr;
  out->n = p->filter_mode == MI355_FILTER_NONE ? 0 : p->n_filter;
  if (out->mode == MI355_FILTER_NONE || out->n == 0) return MI355_OK;
  if (p->io_mem == MI355_MEM_DEVICE) {
    out->ids = p->filter_rowids;
    return MI355_OK;
  }
  ST_TRY(stage.ensure(sizeof(uint64_t) * out->n));
  HIP_TRY(hipMemcpyAsync(stage.p, p->filter_rowids, sizeof(uint64_t) * out->n, hipMemcpyHostToDevice, st));
  out->ids = stage.as<uint64_t>();
  return MI355_OK;
}

static int kpl_for(uint32_t kk) { return kk <= 64 ? 1 : kk <= 128 ? 2 : kk <= 256 ? 4 : 0; }

template <int VPT, int NT>
static int32_t launch_scan_pair_kpl(const ScanArgs& sa, dim3 grid, size_t lds, hipStream_t st,
                                    int kpl) {
#define LAUNCH_SP(K)                                                                       \
  {                                                                                        \
    auto kern = k_scan_pair<VPT, K, NT>;                                                   \
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds));                                                \
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, st, sa);                                 \
  }
  if (kpl == 1) LAUNCH_SP(1)
  else if (kpl == 2) LAUNCH_SP(2)
  else LAUNCH_SP(4)
#undef LAUNCH_SP
  HIP_TRY(hipGetLastError());
  return MI355_OK;
}

static int32_t launch_scan_pair(const ScanArgs& sa, dim3 grid, size_t lds, hipStream_t st, int kpl,
                                uint32_t vpt, uint32_t nt) {
  if (vpt == 4 && nt == 256) return launch_scan_pair_kpl<4, 256>(sa, grid, lds, st, kpl);
  if (vpt == 4 && nt == 512) return launch_scan_pair_kpl<4, 512>(sa, grid, lds, st, kpl);
  if (vpt == 4 && nt == 1024) return launch_scan_pair_kpl<4, 1024>(sa, grid, lds, st, kpl);
  if (vpt == 16 && nt == 256) return launch_scan_pair_kpl<16, 256>(sa, grid, lds, st, kpl);
  if (vpt == 16 && nt == 512) return launch_scan_pair_kpl<16, 512>(sa, grid, lds, st, kpl);
  if (vpt == 16 && nt == 1024) return launch_scan_pair_kpl<16, 1024>(sa, grid, lds, st, kpl);
  return fail(MI355_ERR_INVALID_INPUT, "unsupported scan tuning vpt=%u threads=%u", vpt, nt);
}

template <typename Args, typename KernFn>
static void launch_by_kpl(int kpl, KernFn k1, KernFn k2, KernFn k4, dim3 grid, dim3 block,
                          size_t lds, hipStream_t st, const Args& a) {
  if (kpl == 1)
    hipLaunchKernelGGL(k1, grid, block, lds, st, a);
  else if (kpl == 2)
    hipLaunchKernelGGL(k2, grid, block, lds, st, a);
  else
    hipLaunchKernelGGL(k4, grid, block, lds, st, a);
}

template <int M>
static int32_t launch_scan_skew_m(const SkewArgs& sa, uint32_t n_blocks, uint32_t dim, uint32_t kk,
                                  hipStream_t st) {
  auto lds_of = [&](int nw, int lr) {
    return (size_t)SK_TABLE_BYTES + (((size_t)dim * 4 + 15) & ~(size_t)15) +
           (size_t)nw * lr * 64 * 8 + (size_t)(nw + 10) * 4 + 96;
  };
#define LAUNCH_SK(LR, NT)                                                                       \
  {                                                                                             \
    auto kern = k_scan_skew<M, LR, NT>;                                                         \
    const size_t lds = lds_of(NT / 64, LR);                                                     \
    if (lds > 160u * 1024)                                                                      \
      return fail(MI355_ERR_NOT_SUPPORTED, "scan work item needs %zu B of LDS (> 160 KiB)", lds); \
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                (int)lds));                                                     \
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(NT), lds, st, sa);                            \
  }
  if (kk <= 64) LAUNCH_SK(2, 1024)
  else if (kk <= 128) LAUNCH_SK(3, 1024)
  else LAUNCH_SK(5, 512)
#undef LAUNCH_SK
  HIP_TRY(hipGetLastError());
  return MI355_OK;
}

static int32_t launch_scan_skew(const SkewArgs& sa, uint32_t m, uint32_t n_blocks, uint32_t dim,
                                uint32_t kk, hipStream_t st) {
  switch (m) {
    case 32: return launch_scan_skew_m<32>(sa, n_blocks, dim, kk, st);
    case 48: return launch_scan_skew_m<48>(sa, n_blocks, dim, kk, st);
    case 64: return launch_scan_skew_m<64>(sa, n_blocks, dim, kk, st);
    case 80: return launch_scan_skew_m<80>(sa, n_blocks, dim, kk, st);
    case 96: return launch_scan_skew_m<96>(sa, n_blocks, dim, kk, st);
  }
  return fail(MI355_ERR_NOT_SUPPORTED, "no skewed scan kernel for m = %u", m);
}

struct SearchPlan {
  uint32_t k, kk, nprobe;
  bool refine;
  RangeFilter range;
  RowFilter filter;
  const uint64_t* ext_probes = nullptr;  // device [nq, nprobe]: skip the coarse stage (mi355_search_probes)
};

// one pass of the pipeline over `nq` queries already resident at d_q;
// results land in d_ids/d_dist/d_cnt (device, [nq,k])
// d_cnt_ann [nq]: rows the ANN stage found per query, BEFORE the refine re-rank (what
// maximum_nprobes compares with k * refine_factor); may alias d_cnt when there is no refine
static int32_t run_ivfpq(mi355_index* ix, const float* d_q, uint32_t nq, const SearchPlan& pl,
                         uint64_t* d_ids, float* d_dist, uint32_t* d_cnt, uint32_t* d_cnt_ann) {
  hipStream_t st = ix->stream;
  const IndexView view = make_view(ix);
  const uint32_t nprobe = pl.nprobe;
  const int kpl_kk = kpl_for(pl.kk), kpl_k = kpl_for(pl.k);

  const bool skew = ix->layout == MI355_SCAN_SKEW;
  // tuning (dev knobs; defaults chosen from the index shape)
  uint32_t nt = env_u32("MI355_SCAN_THREADS", 0), vpt = env_u32("MI355_SCAN_VPT", 0);
  if (!nt) nt = pl.kk > 64 ? 256 : ix->max_len >= 8192 ? 1024 : ix->max_len >= 2048 ? 512 : 256;
  if (!vpt) vpt = ix->max_len >= 4 * nt * 4 ? 16 : 4;
  // Generic kernel: one work item per (query, partition) whenever the batch
  // alone fills the chip: the distance table is then built once per pair and
  // skewed partitions cost no empty blocks.  Small batches (latency mode) split
  // partitions into slices so that >= ~4 work items per CU exist.  The skewed
  // kernel always takes whole partitions (its 16 streams are the split).
  uint32_t slice = ix->slice_rows;
  if (!slice) {
    const uint64_t pairs = (uint64_t)nq * nprobe;
    uint32_t want = pairs >= 1024 ? 1u : (uint32_t)((1024 + pairs - 1) / std::max<uint64_t>(pairs, 1));
    slice = std::max((ix->max_len + want - 1) / want, nt * vpt);
  }
  slice = (slice + 15u) & ~15u;
  const uint32_t n_slices = skew ? 1u : std::max(1u, (ix->max_len + slice - 1) / slice);
  // LDS: distance table + residual + per-wave candidate lists + wave counters
  const uint32_t lr = pl.kk <= 64 ? 2 : 5;
  const size_t lds = (size_t)ix->m * 1024 + (((size_t)ix->dim * 4 + 15) & ~(size_t)15) +
                     (size_t)(nt / 64) * lr * 64 * 8 + (size_t)(nt / 64) * 4;
  if (!skew && lds > 160u * 1024)
    return fail(MI355_ERR_NOT_SUPPORTED, "scan work item needs %zu B of LDS (> 160 KiB)", lds);

  // chunk the batch so the workspace stays bounded
  const size_t per_q = (size_t)ix->nlist * 4 + (size_t)nprobe * n_slices * pl.kk * sizeof(Cand);
  const size_t budget = (size_t)env_u32("MI355_WORKSPACE_MB", 2048) << 20;
  uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(nq, budget / std::max<size_t>(per_q, 1)));
  chunk = std::min(chunk, 65535u);  // grid.z limit
  if (skew) {
    ST_TRY(ix->items.ensure(sizeof(SkewItem) * (size_t)chunk * nprobe));
    ST_TRY(ix->qthr.ensure(sizeof(uint32_t) * chunk));
  }
  ST_TRY(ix->w_qp.ensure(sizeof(float) * (size_t)chunk * ix->dim));
  ST_TRY(ix->w_qq.ensure(sizeof(float) * chunk));
  ST_TRY(ix->w_coarse.ensure(sizeof(float) * (size_t)chunk * ix->nlist));
  ST_TRY(ix->w_probes.ensure(sizeof(uint32_t) * (size_t)chunk * nprobe));
  ST_TRY(ix->w_cand.ensure(sizeof(Cand) * (size_t)chunk * nprobe * n_slices * pl.kk));
  if (pl.refine) {
    ST_TRY(ix->w_ids2.ensure(sizeof(uint64_t) * (size_t)chunk * pl.kk));
    ST_TRY(ix->w_dist2.ensure(sizeof(float) * (size_t)chunk * pl.kk));
    ST_TRY(ix->w_pos.ensure(sizeof(uint32_t) * (size_t)chunk * pl.kk));
  }
  unsigned long long* d_stat = ix->w_stat.as<unsigned long long>();
  const bool prof = ix->profile != 0;

  for (uint32_t q0 = 0; q0 < nq; q0 += chunk) {
    const uint32_t n = std::min(chunk, nq - q0);
    const float* q = d_q + (size_t)q0 * ix->dim;
    EventSet es{};
    if (prof) {
      if (!ix->ev_free.empty()) {
        es = ix->ev_free.back();
        ix->ev_free.pop_back();
      } else {
        for (auto& e : es.ev) HIP_TRY(hipEventCreate(&e));
      }
      HIP_TRY(hipEventRecord(es.ev[0], st));
    }
    hipLaunchKernelGGL(k_prep_queries, dim3((n + 3) / 4), dim3(256), 4 * (((size_t)ix->dim + 3) & ~(size_t)3) * 4, st,
                       q, n, ix->dim, ix->metric, ix->w_qp.as<float>(), ix->w_qq.as<float>());
    if (pl.ext_probes) {
      // the probe list came from the two-phase coarse stage
      HIP_TRY(hipMemsetAsync(ix->w_bad.p, 0, 4, st));
      const uint32_t np = n * nprobe;
      hipLaunchKernelGGL(k_take_probes, dim3((np + 255) / 256), dim3(256), 0, st, pl.ext_probes + (size_t)q0 * nprobe, np,
                         ix->nlist, view.plen, ix->w_probes.as<uint32_t>(), d_stat, ix->w_bad.as<uint32_t>());
      HIP_TRY(hipGetLastError());
      if (prof) HIP_TRY(hipEventRecord(es.ev[1], st));
    } else {
    if (env_u32("MI355_COARSE_VALU", 0))  // dev knob: the register-tiled VALU kernel (same bits)
      hipLaunchKernelGGL(k_coarse_tile, dim3((ix->nlist + CO_T - 1) / CO_T, (n + CO_T - 1) / CO_T),
                         dim3(256), 0, st, ix->w_qp.as<float>(), ix->w_qq.as<float>(), n,
                         view.centroids, view.cnorm, ix->nlist, ix->dim, ix->metric,
                         ix->w_coarse.as<float>());
    else
      hipLaunchKernelGGL(k_coarse_mfma, dim3((ix->nlist + CM_T - 1) / CM_T, (n + CM_T - 1) / CM_T),
                         dim3(256), 0, st, ix->w_qp.as<float>(), ix->w_qq.as<float>(), n,
                         view.centroids, view.cnorm, ix->nlist, ix->dim, ix->metric,
                         ix->w_coarse.as<float>());
    HIP_TRY(hipGetLastError());
    if (prof) HIP_TRY(hipEventRecord(es.ev[1], st));
    hipLaunchKernelGGL(k_select_probes, dim3(n), dim3(256), 0, st, ix->w_coarse.as<float>(),
                       ix->nlist, nprobe, view.plen, ix->w_probes.as<uint32_t>(), d_stat);
    HIP_TRY(hipGetLastError());
    }
    if (prof) HIP_TRY(hipEventRecord(es.ev[2], st));

    if (skew) {
      PlanArgs pa;
      pa.probes = ix->w_probes.as<uint32_t>();
      pa.n_pairs = n * nprobe;
      pa.nlist = ix->nlist;
      pa.plen = view.plen;
      pa.order = ix->order.as<uint32_t>();
      pa.xcd_first = ix->xcd_first.as<uint32_t>();
      pa.cnt = ix->p_cnt.as<uint32_t>();
      pa.off = ix->p_off.as<uint32_t>();
      pa.fill = ix->p_fill.as<uint32_t>();
      pa.q_start = ix->q_start.as<uint32_t>();
      pa.heads = ix->heads.as<uint32_t>();
      pa.items = ix->items.as<SkewItem>();
      pa.lrow0 = view.lrow0;
      pa.grow0 = view.grow0;
      pa.code_off = view.code_off;
      pa.cand = ix->w_cand.as<Cand>();
      pa.kk = pl.kk;
      const uint32_t pb = (pa.n_pairs + 255) / 256;
      HIP_TRY(hipMemsetAsync(ix->qthr.p, 0xFF, sizeof(uint32_t) * n, st));
      hipLaunchKernelGGL(k_plan_count, dim3(pb), dim3(256), 0, st, pa);
      hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(1024), 0, st, pa);
      hipLaunchKernelGGL(k_plan_fill, dim3(pb), dim3(256), 0, st, pa);
      HIP_TRY(hipGetLastError());
      SkewArgs ka;
      ka.ix = view;
      ka.cbT = ix->cbT.as<float>();
      ka.qp = ix->w_qp.as<float>();
      ka.probes = ix->w_probes.as<uint32_t>();
      ka.items = ix->items.as<SkewItem>();
      ka.q_start = ix->q_start.as<uint32_t>();
      ka.heads = ix->heads.as<uint32_t>();
      ka.qthr = ix->qthr.as<uint32_t>();
      ka.nprobe = nprobe;
      ka.kk = pl.kk;
      ka.range = pl.range;
      ka.filter = pl.filter;
      ka.cand = ix->w_cand.as<Cand>();
      ka.dbg = env_u32("MI355_DBG_SKIP", 0);
      const uint32_t n_blocks = (uint32_t)std::min<uint64_t>(ix->n_cus, (uint64_t)n * nprobe);
      ST_TRY(launch_scan_skew(ka, ix->m, std::max(n_blocks, 1u), ix->dim, pl.kk, st));
    } else {
      ScanArgs sa;
      sa.ix = view;
      sa.qp = ix->w_qp.as<float>();
      sa.probes = ix->w_probes.as<uint32_t>();
      sa.nprobe = nprobe;
      sa.slice_rows = slice;
      sa.n_slices = n_slices;
      sa.kk = pl.kk;
      sa.range = pl.range;
      sa.filter = pl.filter;
      sa.cand = ix->w_cand.as<Cand>();
      sa.dbg = env_u32("MI355_DBG_SKIP", 0);
      ST_TRY(launch_scan_pair(sa, dim3(n_slices, nprobe, n), lds, st, kpl_kk, vpt, nt));
    }
    if (prof) HIP_TRY(hipEventRecord(es.ev[3], st));

    MergeArgs ma;
    ma.cand = ix->w_cand.as<Cand>();
    ma.n_src = nprobe * n_slices;
    ma.kk_in = pl.kk;
    if (!pl.refine) {
      ma.k_out = pl.k;
      ma.out_ids = d_ids + (size_t)q0 * pl.k;
      ma.out_dist = d_dist + (size_t)q0 * pl.k;
      ma.out_pos = nullptr;
      ma.out_cnt = d_cnt + q0;
      launch_by_kpl(kpl_k, k_merge_cands<1>, k_merge_cands<2>, k_merge_cands<4>, dim3(n), dim3(64), 0, st, ma);
      HIP_TRY(hipGetLastError());
      if (prof) HIP_TRY(hipEventRecord(es.ev[4], st));
    } else {
      ma.k_out = pl.kk;
      ma.out_ids = ix->w_ids2.as<uint64_t>();
      ma.out_dist = ix->w_dist2.as<float>();
      ma.out_pos = ix->w_pos.as<uint32_t>();
      ma.out_cnt = d_cnt_ann + q0;
      launch_by_kpl(kpl_kk, k_merge_cands<1>, k_merge_cands<2>, k_merge_cands<4>, dim3(n), dim3(64), 0, st, ma);
      HIP_TRY(hipGetLastError());
      if (prof) HIP_TRY(hipEventRecord(es.ev[4], st));
      RefineArgs ra;
      ra.ix = view;
      ra.q = q;
      ra.in_ids = ix->w_ids2.as<uint64_t>();
      ra.in_pos = ix->w_pos.as<uint32_t>();
      ra.in_cnt = d_cnt_ann + q0;
      ra.kk = pl.kk;
      ra.k = pl.k;
      ra.range = pl.range;
      ra.out_ids = d_ids + (size_t)q0 * pl.k;
      ra.out_dist = d_dist + (size_t)q0 * pl.k;
      ra.out_cnt = d_cnt + q0;
      size_t rl = (((size_t)ix->dim * 4 + 15) & ~(size_t)15) + sizeof(Cand) * pl.kk;
      launch_by_kpl(kpl_k, k_refine<1>, k_refine<2>, k_refine<4>, dim3(n), dim3(256), rl, st, ra);
      HIP_TRY(hipGetLastError());
    }
    if (prof) {
      HIP_TRY(hipEventRecord(es.ev[5], st));
      ix->ev_pending.push_back(es);
    }
  }
  ix->stats.n_queries += nq;
  ix->stats.work_items += (uint64_t)nq * nprobe * n_slices;
  ix->stats.partitions_probed += (uint64_t)nq * nprobe;
  ix->stats.scan_variant = ix->layout;
  return MI355_OK;
}

// ext_probes != NULL: mi355_search_probes (the probe list replaces the coarse stage)
static int32_t search_impl(mi355_index* ix, const float* queries, uint32_t n_queries,
                           const mi355_search_params* p, const uint64_t* ext_probes, uint32_t ext_nprobe,
                           uint64_t* out_rowids, float* out_dist, uint32_t* out_counts) {
  if (!ix) return fail(MI355_ERR_INVALID_INPUT, "index is NULL");
  ST_TRY(validate_params(p));
  // nprobes validation: rust/lancedb/src/query.rs:1232-1275
  if (!ext_probes) {
    if (p->nprobe_min == 0) return fail(MI355_ERR_INVALID_INPUT, "minimum_nprobes must be greater than 0");
    if (p->nprobe_max != 0 && p->nprobe_max < p->nprobe_min)
      return fail(MI355_ERR_INVALID_INPUT, "maximum_nprobes must be greater than or equal to minimum_nprobes");
  } else if (ext_nprobe == 0 || ext_nprobe > ix->nlist) {
    return fail(MI355_ERR_INVALID_INPUT, "probe list length %u must be in 1..nlist (%u)", ext_nprobe, ix->nlist);
  }
  if (p->metric != MI355_METRIC_DEFAULT && p->metric != ix->metric)
    return fail(MI355_ERR_INVALID_INPUT,
                "distance type %u does not match the metric the index was trained with (%u)",
                p->metric, ix->metric);
  if (n_queries == 0) return MI355_OK;
  if (!queries || !out_counts || (p->k && (!out_rowids || !out_dist)))
    return fail(MI355_ERR_INVALID_INPUT, "NULL query / output buffer");
  if (p->refine_factor && !ix->has_raw)
    return fail(MI355_ERR_INVALID_INPUT, "refine_factor needs raw vectors on the index handle");
  const uint32_t k = p->k;
  uint64_t kk64 = (uint64_t)k * (p->refine_factor ? p->refine_factor : 1);
  if (k == 0) {
    if (p->io_mem == MI355_MEM_HOST) memset(out_counts, 0, sizeof(uint32_t) * n_queries);
    else {
      HIP_TRY(hipSetDevice(ix->device));
      HIP_TRY(hipMemsetAsync(out_counts, 0, sizeof(uint32_t) * n_queries, ix->stream));
    }
    return MI355_OK;
  }
  if (kk64 > 256 || kpl_for((uint32_t)kk64) == 0)
    return fail(MI355_ERR_NOT_SUPPORTED, "k * refine_factor = %llu exceeds the supported 256",
                (unsigned long long)kk64);
  uint32_t np_min = std::min(p->nprobe_min, ix->nlist);
  uint32_t np_max = (p->nprobe_max == 0 || p->nprobe_max > ix->nlist) ? ix->nlist : p->nprobe_max;
  if (ext_probes) np_min = np_max = ext_nprobe;
  if (ix->shard_count > 1 && np_max != np_min)
    return fail(MI355_ERR_NOT_SUPPORTED,
                "maximum_nprobes expansion on a sharded handle must be driven by the caller after "
                "the cross-shard merge");

  std::lock_guard<std::mutex> lk(ix->mu);
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t st = ix->stream;
  auto t_start = std::chrono::steady_clock::now();
  if (ix->profile != 2) {  // 2 = cumulative: counters run until the next configure()
    ST_TRY(drain_events(ix, true));
    reset_stats(ix);
    HIP_TRY(hipMemsetAsync(ix->w_stat.p, 0, 64, st));
  }

  const bool host_io = p->io_mem == MI355_MEM_HOST;
  const float* d_q = queries;
  uint64_t* d_ids = out_rowids;
  float* d_dist = out_dist;
  uint32_t* d_cnt = out_counts;
  if (host_io) {
    ST_TRY(ix->w_q.ensure(sizeof(float) * (size_t)n_queries * ix->dim));
    ST_TRY(ix->w_ids.ensure(sizeof(uint64_t) * (size_t)n_queries * k));
    ST_TRY(ix->w_dist.ensure(sizeof(float) * (size_t)n_queries * k));
    ST_TRY(ix->w_cnt.ensure(sizeof(uint32_t) * n_queries));
    HIP_TRY(hipMemcpyAsync(ix->w_q.p, queries, sizeof(float) * (size_t)n_queries * ix->dim, hipMemcpyHostToDevice, st));
    d_q = ix->w_q.as<float>();
    d_ids = ix->w_ids.as<uint64_t>();
    d_dist = ix->w_dist.as<float>();
    d_cnt = ix->w_cnt.as<uint32_t>();
  }
  SearchPlan pl;
  pl.k = k;
  pl.kk = (uint32_t)kk64;
  pl.refine = p->refine_factor != 0;
  pl.nprobe = np_min;
  pl.range.has_lower = p->has_lower_bound;
  pl.range.has_upper = p->has_upper_bound;
  pl.range.lower = p->lower_bound;
  pl.range.upper = p->upper_bound;
  ST_TRY(make_row_filter(p, ix->w_filter, st, &pl.filter));
  if (ext_probes) {
    pl.ext_probes = ext_probes;
    if (host_io) {
      const size_t pb = sizeof(uint64_t) * (size_t)n_queries * ext_nprobe;
      ST_TRY(ix->w_probes64.ensure(pb));
      HIP_TRY(hipMemcpyAsync(ix->w_probes64.p, ext_probes, pb, hipMemcpyHostToDevice, st));
      pl.ext_probes = ix->w_probes64.as<uint64_t>();
    }
  }
  uint32_t* d_cnt_ann = d_cnt;
  if (pl.refine) {
    ST_TRY(ix->w_cnt2.ensure(sizeof(uint32_t) * n_queries));
    d_cnt_ann = ix->w_cnt2.as<uint32_t>();
  }
  ST_TRY(run_ivfpq(ix, d_q, n_queries, pl, d_ids, d_dist, d_cnt, d_cnt_ann));

  if (np_max > np_min) {
    // maximum_nprobes (query.rs:1246-1262): queries whose ANN stage found fewer than the
    // k * refine_factor rows it was asked for are searched again over the first np_max
    // partitions (the decision is taken before the refine re-rank, as in the oracle).
    std::vector<uint32_t> cnt(n_queries);
    HIP_TRY(hipMemcpyAsync(cnt.data(), d_cnt_ann, sizeof(uint32_t) * n_queries, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<uint32_t> shortq;
    for (uint32_t i = 0; i < n_queries; ++i)
      if (cnt[i] < pl.kk) shortq.push_back(i);
    if (!shortq.empty()) {
      const uint32_t ns = (uint32_t)shortq.size();
      DevBuf sq, sids, sdist, scnt, scnt_ann;
      ST_TRY(sq.ensure(sizeof(float) * (size_t)ns * ix->dim));
      ST_TRY(sids.ensure(sizeof(uint64_t) * (size_t)ns * k));
      ST_TRY(sdist.ensure(sizeof(float) * (size_t)ns * k));
      ST_TRY(scnt.ensure(sizeof(uint32_t) * ns));
      ST_TRY(scnt_ann.ensure(sizeof(uint32_t) * ns));
      for (uint32_t i = 0; i < ns; ++i)
        HIP_TRY(hipMemcpyAsync(sq.as<float>() + (size_t)i * ix->dim, d_q + (size_t)shortq[i] * ix->dim,
                               sizeof(float) * ix->dim, hipMemcpyDeviceToDevice, st));
      SearchPlan p2 = pl;
      p2.nprobe = np_max;
      int32_t s2 = run_ivfpq(ix, sq.as<float>(), ns, p2, sids.as<uint64_t>(), sdist.as<float>(), scnt.as<uint32_t>(),
                             pl.refine ? scnt_ann.as<uint32_t>() : scnt.as<uint32_t>());
      if (s2 == MI355_OK) {
        for (uint32_t i = 0; i < ns && s2 == MI355_OK; ++i) {
          size_t o = (size_t)shortq[i] * k;
          if (hipMemcpyAsync(d_ids + o, sids.as<uint64_t>() + (size_t)i * k, sizeof(uint64_t) * k, hipMemcpyDeviceToDevice, st) != hipSuccess ||
              hipMemcpyAsync(d_dist + o, sdist.as<float>() + (size_t)i * k, sizeof(float) * k, hipMemcpyDeviceToDevice, st) != hipSuccess ||
              hipMemcpyAsync(d_cnt + shortq[i], scnt.as<uint32_t>() + i, sizeof(uint32_t), hipMemcpyDeviceToDevice, st) != hipSuccess)
            s2 = fail(MI355_ERR_RUNTIME, "scatter of expanded-probe results failed");
        }
      }
      (void)hipStreamSynchronize(st);
      sq.release();
      sids.release();
      sdist.release();
      scnt.release();
      scnt_ann.release();
      if (s2 != MI355_OK) return s2;
    }
  }

  if (ext_probes && host_io) {  // ids outside 0..nlist-1 are a caller error: report instead of scanning garbage
    uint32_t bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, ix->w_bad.p, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (bad) return fail(MI355_ERR_INVALID_INPUT, "%u probe ids are not partitions of this index", bad);
  }
  if (host_io) {
    HIP_TRY(hipMemcpyAsync(out_rowids, d_ids, sizeof(uint64_t) * (size_t)n_queries * k, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_dist, d_dist, sizeof(float) * (size_t)n_queries * k, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_counts, d_cnt, sizeof(uint32_t) * n_queries, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (p->timeout_ms) {
      auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_start).count();
      if (ms > (long long)p->timeout_ms)
        return fail(MI355_ERR_TIMEOUT, "Query timeout: %lld ms > %u ms", (long long)ms, p->timeout_ms);
    }
  }
  return MI355_OK;
}

extern "C" int32_t mi355_search(mi355_index* ix, const float* queries, uint32_t n_queries,
                                const mi355_search_params* p, uint64_t* out_rowids,
                                float* out_dist, uint32_t* out_counts) {
  return search_impl(ix, queries, n_queries, p, nullptr, 0, out_rowids, out_dist, out_counts);
}

extern "C" int32_t mi355_search_probes(mi355_index* ix, const float* queries, uint32_t n_queries,
                                       const mi355_search_params* p, const uint64_t* probes, uint32_t nprobe,
                                       uint64_t* out_rowids, float* out_dist, uint32_t* out_counts) {
  if (!probes) return fail(MI355_ERR_INVALID_INPUT, "probes is NULL");
  return search_impl(ix, queries, n_queries, p, probes, nprobe, out_rowids, out_dist, out_counts);
}

extern "C" int32_t mi355_coarse_topn(mi355_index* ix, const float* queries, uint32_t n_queries, uint32_t nprobe,
                                     uint32_t cent_lo, uint32_t cent_hi, uint32_t io_mem, uint64_t* out_part_ids,
                                     float* out_dist, uint32_t* out_counts) {
  if (!ix) return fail(MI355_ERR_INVALID_INPUT, "index is NULL");
  if (io_mem > MI355_MEM_DEVICE) return fail(MI355_ERR_INVALID_INPUT, "bad io_mem");
  if (cent_lo >= cent_hi || cent_hi > ix->nlist)
    return fail(MI355_ERR_INVALID_INPUT, "centroid slice [%u, %u) is not inside 0..%u", cent_lo, cent_hi, ix->nlist);
  if (nprobe == 0 || nprobe > 256) return fail(MI355_ERR_INVALID_INPUT, "nprobe must be in 1..256 for the two-phase search");
  if (n_queries == 0) return MI355_OK;
  if (!queries || !out_part_ids || !out_dist || !out_counts) return fail(MI355_ERR_INVALID_INPUT, "NULL buffer");
  std::lock_guard<std::mutex> lk(ix->mu);
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t st = ix->stream;
  const bool host_io = io_mem == MI355_MEM_HOST;
  const uint32_t n_slice = cent_hi - cent_lo, n_sel = std::min(nprobe, n_slice), nq = n_queries;
  const float* d_q = queries;
  uint64_t* d_ids = out_part_ids;
  float* d_dist = out_dist;
  uint32_t* d_cnt = out_counts;
  if (host_io) {
    ST_TRY(ix->w_q.ensure(sizeof(float) * (size_t)nq * ix->dim));
    ST_TRY(ix->w_ids.ensure(sizeof(uint64_t) * (size_t)nq * nprobe));
    ST_TRY(ix->w_dist.ensure(sizeof(float) * (size_t)nq * nprobe));
    ST_TRY(ix->w_cnt.ensure(sizeof(uint32_t) * nq));
    HIP_TRY(hipMemcpyAsync(ix->w_q.p, queries, sizeof(float) * (size_t)nq * ix->dim, hipMemcpyHostToDevice, st));
    d_q = ix->w_q.as<float>();
    d_ids = ix->w_ids.as<uint64_t>();
    d_dist = ix->w_dist.as<float>();
    d_cnt = ix->w_cnt.as<uint32_t>();
  }
  ST_TRY(ix->w_qp.ensure(sizeof(float) * (size_t)nq * ix->dim));
  ST_TRY(ix->w_qq.ensure(sizeof(float) * nq));
  ST_TRY(ix->w_coarse.ensure(sizeof(float) * (size_t)nq * n_slice));
  ST_TRY(ix->w_probes.ensure(sizeof(uint32_t) * (size_t)nq * std::max(n_sel, 1u)));
  hipLaunchKernelGGL(k_prep_queries, dim3((nq + 3) / 4), dim3(256), 4 * (((size_t)ix->dim + 3) & ~(size_t)3) * 4, st,
                     d_q, nq, ix->dim, ix->metric, ix->w_qp.as<float>(), ix->w_qq.as<float>());
  // the slice's centroids, norms and partition lengths are contiguous sub-ranges of the handle's arrays
  const float* cen = ix->centroids.as<float>() + (size_t)cent_lo * ix->dim;
  const float* cn = ix->cnorm.as<float>() + cent_lo;
  for (uint32_t y0 = 0; y0 < nq; y0 += 65535u * CM_T) {  // grid.y limit
    const uint32_t ny = std::min(nq - y0, 65535u * CM_T);
    hipLaunchKernelGGL(k_coarse_mfma, dim3((n_slice + CM_T - 1) / CM_T, (ny + CM_T - 1) / CM_T), dim3(256), 0, st,
                       ix->w_qp.as<float>() + (size_t)y0 * ix->dim, ix->w_qq.as<float>() + y0, ny, cen, cn, n_slice,
                       ix->dim, ix->metric, ix->w_coarse.as<float>() + (size_t)y0 * n_slice);
  }
  hipLaunchKernelGGL(k_select_probes, dim3(nq), dim3(256), 0, st, ix->w_coarse.as<float>(), n_slice, n_sel,
                     ix->plen.as<uint32_t>() + cent_lo, ix->w_probes.as<uint32_t>(), (unsigned long long*)nullptr);
  const uint32_t np = nq * nprobe;
  hipLaunchKernelGGL(k_emit_coarse_pairs, dim3((np + 255) / 256), dim3(256), 0, st, ix->w_probes.as<uint32_t>(),
                     ix->w_coarse.as<float>(), nq, n_sel, n_slice, nprobe, cent_lo, d_ids, d_dist, d_cnt);
  HIP_TRY(hipGetLastError());
  if (host_io) {
    HIP_TRY(hipMemcpyAsync(out_part_ids, d_ids, sizeof(uint64_t) * (size_t)nq * nprobe, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_dist, d_dist, sizeof(float) * (size_t)nq * nprobe, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_counts, d_cnt, sizeof(uint32_t) * nq, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  return MI355_OK;
}

// ------------------------------------------------------------------ encode --
namespace {
struct EncodeScratch {  // released on every exit path
  DevBuf cen, cb, cn, x, qp, qq, coarse, assign, hist, codes_src, codes_dst, order, cntB, lrank, run[2];
  hipStream_t st = nullptr;
  ~EncodeScratch() {
    for (DevBuf* b : {&cen, &cb, &cn, &x, &qp, &qq, &coarse, &assign, &hist, &codes_src, &codes_dst, &order, &cntB,
                      &lrank, &run[0], &run[1]})
      b->release();
    if (st) (void)hipStreamDestroy(st);
  }
};
}  // namespace

// stable counting sort of the rows by partition: order[position] = source row (chunks of
// <= 65536 rows = 256 blocks; `base` = exclusive prefix sums of the partition histogram)
static int32_t stable_order(hipStream_t st, const uint32_t* d_assign, uint64_t n_rows, uint32_t nlist,
                            const std::vector<unsigned long long>& base, DevBuf& cntB, DevBuf& lrank, DevBuf* run,
                            uint64_t* d_order) {
  const uint64_t bchunk = 65536;
  const uint32_t max_blocks = (uint32_t)((std::min(bchunk, n_rows) + 255) / 256);
  ST_TRY(cntB.ensure(sizeof(uint32_t) * (size_t)std::max(max_blocks, 1u) * nlist));
  ST_TRY(lrank.ensure(sizeof(uint32_t) * (size_t)std::max(max_blocks, 1u) * 256));
  ST_TRY(run[0].ensure(sizeof(unsigned long long) * nlist));
  ST_TRY(run[1].ensure(sizeof(unsigned long long) * nlist));
  HIP_TRY(hipMemcpyAsync(run[0].p, base.data(), sizeof(unsigned long long) * nlist, hipMemcpyHostToDevice, st));
  int cur = 0;
  for (uint64_t r0 = 0; r0 < n_rows; r0 += bchunk, cur ^= 1) {
    const uint32_t nb = (uint32_t)((std::min(bchunk, n_rows - r0) + 255) / 256);
    HIP_TRY(hipMemsetAsync(cntB.p, 0, sizeof(uint32_t) * (size_t)nb * nlist, st));
    hipLaunchKernelGGL(k_local_rank, dim3(nb), dim3(256), 0, st, d_assign, r0, n_rows, nlist, cntB.as<uint32_t>(),
                       lrank.as<uint32_t>());
    hipLaunchKernelGGL(k_block_scan, dim3((nlist + 255) / 256), dim3(256), 0, st, cntB.as<uint32_t>(), nb, nlist,
                       run[cur].as<unsigned long long>(), run[cur ^ 1].as<unsigned long long>());
    hipLaunchKernelGGL(k_positions, dim3(nb), dim3(256), 0, st, d_assign, r0, n_rows, nlist, cntB.as<uint32_t>(),
                       lrank.as<uint32_t>(), run[cur].as<unsigned long long>(), d_order);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));  // `base` (pageable host memory) is still being read by the upload
  return MI355_OK;
}

extern "C" int32_t mi355_ivfpq_encode(const mi355_encode_desc* d, const float* vectors, uint64_t n_rows,
                                      uint64_t* out_part_offsets, uint8_t* out_codes, uint64_t* out_order,
                                      uint32_t* out_assign) {
  if (!d) return fail(MI355_ERR_INVALID_INPUT, "desc is NULL");
  if (d->struct_size != sizeof(mi355_encode_desc))
    return fail(MI355_ERR_INVALID_INPUT, "mi355_encode_desc.struct_size %u != %zu (ABI mismatch)", d->struct_size,
                sizeof(mi355_encode_desc));
  if (d->dim == 0 || d->nlist == 0 || d->m == 0) return fail(MI355_ERR_INVALID_INPUT, "dim, nlist and m must be > 0");
  if (d->dim % d->m) return fail(MI355_ERR_INVALID_INPUT, "dim %u is not a multiple of m %u", d->dim, d->m);
  if (d->nbits != 8) return fail(MI355_ERR_NOT_SUPPORTED, "only 8-bit PQ codes (nbits = %u)", d->nbits);
  if (d->metric > MI355_METRIC_DOT || d->mem > MI355_MEM_DEVICE) return fail(MI355_ERR_INVALID_INPUT, "bad metric / mem enum");
  if (!d->centroids || !d->codebook || !out_part_offsets) return fail(MI355_ERR_INVALID_INPUT, "NULL buffer");
  if (n_rows && (!vectors || !out_codes || !out_order)) return fail(MI355_ERR_INVALID_INPUT, "NULL buffer");
  const uint32_t dim = d->dim, nlist = d->nlist, m = d->m, dsub = dim / m;
  const uint32_t jt = (m % 4 == 0) ? 4 : 1;
  if ((size_t)jt * 256 * dsub * 4 > 150u * 1024)
    return fail(MI355_ERR_NOT_SUPPORTED, "dim / m = %u: the codebook slices do not fit LDS", dsub);
  if ((size_t)dim * 16 > 150u * 1024) return fail(MI355_ERR_NOT_SUPPORTED, "dim %u too large", dim);
  ST_TRY(need_device(d->device));
  if (n_rows == 0) {
    for (uint32_t p = 0; p <= nlist; ++p) out_part_offsets[p] = 0;
    return MI355_OK;
  }
  EncodeScratch w;
  HIP_TRY(hipStreamCreateWithFlags(&w.st, hipStreamNonBlocking));
  hipStream_t st = w.st;
  const bool host = d->mem == MI355_MEM_HOST;
  // chunk of rows: the [chunk, nlist] coarse matrix stays within 1 GiB
  uint64_t chunk = std::min<uint64_t>(65536, ((size_t)1 << 30) / ((size_t)nlist * 4));
  chunk = std::max<uint64_t>(256, chunk & ~(uint64_t)255);
  chunk = std::min<uint64_t>(chunk, (n_rows + 255) & ~(uint64_t)255);

  ST_TRY(w.cen.ensure(sizeof(float) * (size_t)nlist * dim));
  ST_TRY(w.cb.ensure(sizeof(float) * (size_t)m * 256 * dsub));
  ST_TRY(w.cn.ensure(sizeof(float) * nlist));
  ST_TRY(w.qp.ensure(sizeof(float) * chunk * dim));
  ST_TRY(w.qq.ensure(sizeof(float) * chunk));
  ST_TRY(w.coarse.ensure(sizeof(float) * chunk * nlist));
  ST_TRY(w.assign.ensure(sizeof(uint32_t) * n_rows));
  ST_TRY(w.hist.ensure(sizeof(uint32_t) * nlist));
  ST_TRY(w.codes_src.ensure((size_t)n_rows * m));
  if (host) {
    ST_TRY(w.x.ensure(sizeof(float) * chunk * dim));
    ST_TRY(w.codes_dst.ensure((size_t)n_rows * m));
    ST_TRY(w.order.ensure(sizeof(uint64_t) * n_rows));
  }
  HIP_TRY(copy_in(w.cen.p, d->centroids, sizeof(float) * (size_t)nlist * dim, d->mem, st));
  HIP_TRY(copy_in(w.cb.p, d->codebook, sizeof(float) * (size_t)m * 256 * dsub, d->mem, st));
  hipLaunchKernelGGL(k_centroid_norms, dim3((nlist + 63) / 64), dim3(64), 0, st, w.cen.as<float>(), nlist, dim,
                     w.cn.as<float>());
  HIP_TRY(hipMemsetAsync(w.hist.p, 0, sizeof(uint32_t) * nlist, st));

  // ---- pass A: partition + codes of every row, in source order
  const size_t enc_lds = (size_t)jt * 256 * dsub * 4;
  if (jt == 4)
    HIP_TRY(hipFuncSetAttribute((const void*)k_encode_rows<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_lds));
  else
    HIP_TRY(hipFuncSetAttribute((const void*)k_encode_rows<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_lds));
  for (uint64_t r0 = 0; r0 < n_rows; r0 += chunk) {
    const uint32_t n = (uint32_t)std::min<uint64_t>(chunk, n_rows - r0);
    const float* d_x = vectors + (size_t)r0 * dim;
    if (host) {
      HIP_TRY(hipMemcpyAsync(w.x.p, d_x, sizeof(float) * (size_t)n * dim, hipMemcpyHostToDevice, st));
      d_x = w.x.as<float>();
    }
    hipLaunchKernelGGL(k_prep_queries, dim3((n + 3) / 4), dim3(256), 4 * (((size_t)dim + 3) & ~(size_t)3) * 4, st, d_x,
                       n, dim, d->metric, w.qp.as<float>(), w.qq.as<float>());
    hipLaunchKernelGGL(k_coarse_mfma, dim3((nlist + CM_T - 1) / CM_T, (n + CM_T - 1) / CM_T), dim3(256), 0, st,
                       w.qp.as<float>(), w.qq.as<float>(), n, w.cen.as<float>(), w.cn.as<float>(), nlist, dim,
                       d->metric, w.coarse.as<float>());
    hipLaunchKernelGGL(k_argmin_rows, dim3(n), dim3(256), 0, st, w.coarse.as<float>(), n, nlist,
                       w.assign.as<uint32_t>() + r0, w.hist.as<uint32_t>());
    EncodeArgs ea{w.qp.as<float>(), r0, n, w.assign.as<uint32_t>(), w.cen.as<float>(), w.cb.as<float>(),
                  dim, m, dsub, d->metric, w.codes_src.as<uint8_t>()};
    if (jt == 4)
      hipLaunchKernelGGL(k_encode_rows<4>, dim3((n + 255) / 256, m / 4), dim3(256), enc_lds, st, ea);
    else
      hipLaunchKernelGGL(k_encode_rows<1>, dim3((n + 255) / 256, m), dim3(256), enc_lds, st, ea);
    HIP_TRY(hipGetLastError());
    if (host) HIP_TRY(hipStreamSynchronize(st));  // w.x is reused by the next chunk
  }

  // ---- partition offsets (nlist values: host scan)
  std::vector<uint32_t> hist(nlist);
  HIP_TRY(hipMemcpyAsync(hist.data(), w.hist.p, sizeof(uint32_t) * nlist, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  std::vector<unsigned long long> base(nlist);
  uint64_t run = 0;
  for (uint32_t p = 0; p < nlist; ++p) {
    out_part_offsets[p] = run;
    base[p] = run;
    run += hist[p];
  }
  out_part_offsets[nlist] = run;
  if (run != n_rows) return fail(MI355_ERR_RUNTIME, "partition histogram counts %llu of %llu rows",
                                 (unsigned long long)run, (unsigned long long)n_rows);

  // ---- pass B: stable position of every row
  uint64_t* d_order = host ? w.order.as<uint64_t>() : out_order;
  ST_TRY(stable_order(st, w.assign.as<uint32_t>(), n_rows, nlist, base, w.cntB, w.lrank, w.run, d_order));

  // ---- pass C: code rows into index order
  uint8_t* d_codes = host ? w.codes_dst.as<uint8_t>() : out_codes;
  {
    const uint64_t items = n_rows * (m / jt);
    const uint64_t blocks = (items + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return fail(MI355_ERR_NOT_SUPPORTED, "too many rows for one permute launch");
    if (jt == 4)
      hipLaunchKernelGGL(k_permute_codes<4>, dim3((uint32_t)blocks), dim3(256), 0, st, w.codes_src.as<uint8_t>(),
                         d_order, n_rows, m, d_codes);
    else
      hipLaunchKernelGGL(k_permute_codes<1>, dim3((uint32_t)blocks), dim3(256), 0, st, w.codes_src.as<uint8_t>(),
                         d_order, n_rows, m, d_codes);
    HIP_TRY(hipGetLastError());
  }
  if (host) {
    HIP_TRY(hipMemcpyAsync(out_codes, d_codes, (size_t)n_rows * m, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_order, d_order, sizeof(uint64_t) * n_rows, hipMemcpyDeviceToHost, st));
  }
  if (out_assign)
    HIP_TRY(hipMemcpyAsync(out_assign, w.assign.p, sizeof(uint32_t) * n_rows,
                           host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));
  return MI355_OK;
}

// ---------------------------------------------------------------- training --
namespace {
struct TrainScratch {
  DevBuf raw, xp, qq, cen, cn, coarse, assign, hist, order, po, cntB, lrank, run[2], out;
  hipStream_t st = nullptr;
  ~TrainScratch() {
    for (DevBuf* b : {&raw, &xp, &qq, &cen, &cn, &coarse, &assign, &hist, &order, &po, &cntB, &lrank, &run[0], &run[1],
                      &out})
      b->release();
    if (st) (void)hipStreamDestroy(st);
  }
};
}  // namespace

static int32_t check_kmeans_desc(const mi355_kmeans_desc* d) {
  if (!d) return fail(MI355_ERR_INVALID_INPUT, "desc is NULL");
  if (d->struct_size != sizeof(mi355_kmeans_desc))
    return fail(MI355_ERR_INVALID_INPUT, "mi355_kmeans_desc.struct_size %u != %zu (ABI mismatch)", d->struct_size,
                sizeof(mi355_kmeans_desc));
  if (d->dim == 0 || d->k == 0) return fail(MI355_ERR_INVALID_INPUT, "dim and k must be > 0");
  if (d->metric > MI355_METRIC_DOT || d->mem > MI355_MEM_DEVICE) return fail(MI355_ERR_INVALID_INPUT, "bad metric / mem enum");
  if (d->ld && d->ld < d->dim) return fail(MI355_ERR_INVALID_INPUT, "ld %llu < dim %u", (unsigned long long)d->ld, d->dim);
  if ((size_t)d->dim * 16 > 150u * 1024) return fail(MI355_ERR_NOT_SUPPORTED, "dim %u too large", d->dim);
  return MI355_OK;
}

// rows -> dense, normalised (cosine) device copy + their squared norms
static int32_t train_load_rows(TrainScratch& w, const mi355_kmeans_desc* d, const float* vectors, uint64_t n) {
  hipStream_t st = w.st;
  const uint32_t dim = d->dim;
  const uint64_t ld = d->ld ? d->ld : dim;
  const float* d_raw = vectors;
  if (d->mem == MI355_MEM_HOST || ld != dim) {
    ST_TRY(w.raw.ensure(sizeof(float) * n * dim));
    HIP_TRY(hipMemcpy2DAsync(w.raw.p, sizeof(float) * dim, vectors, sizeof(float) * ld, sizeof(float) * dim, n,
                             d->mem == MI355_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
    d_raw = w.raw.as<float>();
  }
  ST_TRY(w.xp.ensure(sizeof(float) * n * dim));
  ST_TRY(w.qq.ensure(sizeof(float) * n));
  const uint64_t step = 1u << 24;
  for (uint64_t r0 = 0; r0 < n; r0 += step) {
    const uint32_t c = (uint32_t)std::min<uint64_t>(step, n - r0);
    hipLaunchKernelGGL(k_prep_queries, dim3((c + 3) / 4), dim3(256), 4 * (((size_t)dim + 3) & ~(size_t)3) * 4, st,
                       d_raw + (size_t)r0 * dim, c, dim, d->metric, w.xp.as<float>() + (size_t)r0 * dim,
                       w.qq.as<float>() + r0);
  }
  HIP_TRY(hipGetLastError());
  return MI355_OK;
}

// partition of every prepared row + histogram (w.cen holds the centroids)
static int32_t train_assign(TrainScratch& w, const mi355_kmeans_desc* d, uint64_t n) {
  hipStream_t st = w.st;
  const uint32_t dim = d->dim, k = d->k;
  uint64_t chunk = std::min<uint64_t>(65536, ((size_t)1 << 30) / ((size_t)k * 4));
  chunk = std::max<uint64_t>(256, chunk & ~(uint64_t)255);
  chunk = std::min<uint64_t>(chunk, (n + 255) & ~(uint64_t)255);
  ST_TRY(w.cn.ensure(sizeof(float) * k));
  ST_TRY(w.coarse.ensure(sizeof(float) * chunk * k));
  ST_TRY(w.assign.ensure(sizeof(uint32_t) * std::max<uint64_t>(n, 1)));
  ST_TRY(w.hist.ensure(sizeof(uint32_t) * k));
  hipLaunchKernelGGL(k_centroid_norms, dim3((k + 63) / 64), dim3(64), 0, st, w.cen.as<float>(), k, dim, w.cn.as<float>());
  HIP_TRY(hipMemsetAsync(w.hist.p, 0, sizeof(uint32_t) * k, st));
  for (uint64_t r0 = 0; r0 < n; r0 += chunk) {
    const uint32_t c = (uint32_t)std::min<uint64_t>(chunk, n - r0);
    hipLaunchKernelGGL(k_coarse_mfma, dim3((k + CM_T - 1) / CM_T, (c + CM_T - 1) / CM_T), dim3(256), 0, st,
                       w.xp.as<float>() + (size_t)r0 * dim, w.qq.as<float>() + r0, c, w.cen.as<float>(),
                       w.cn.as<float>(), k, dim, d->metric, w.coarse.as<float>());
    hipLaunchKernelGGL(k_argmin_rows, dim3(c), dim3(256), 0, st, w.coarse.as<float>(), c, k,
                       w.assign.as<uint32_t>() + r0, w.hist.as<uint32_t>());
  }
  HIP_TRY(hipGetLastError());
  return MI355_OK;
}

extern "C" int32_t mi355_kmeans_train(const mi355_kmeans_desc* d, const float* vectors, uint64_t n_rows,
                                      float* centroids, uint64_t* out_counts) {
  ST_TRY(check_kmeans_desc(d));
  if (!centroids || (n_rows && !vectors)) return fail(MI355_ERR_INVALID_INPUT, "NULL buffer");
  if (n_rows >> 40) return fail(MI355_ERR_NOT_SUPPORTED, "too many training rows");
  ST_TRY(need_device(d->device));
  const uint32_t dim = d->dim, k = d->k;
  const bool host = d->mem == MI355_MEM_HOST;
  TrainScratch w;
  HIP_TRY(hipStreamCreateWithFlags(&w.st, hipStreamNonBlocking));
  hipStream_t st = w.st;
  ST_TRY(w.cen.ensure(sizeof(float) * (size_t)k * dim));
  HIP_TRY(copy_in(w.cen.p, centroids, sizeof(float) * (size_t)k * dim, d->mem, st));
  std::vector<uint32_t> hist(k, 0);
  if (n_rows && d->iters) {
    ST_TRY(train_load_rows(w, d, vectors, n_rows));
    ST_TRY(w.order.ensure(sizeof(uint64_t) * n_rows));
    ST_TRY(w.po.ensure(sizeof(unsigned long long) * ((size_t)k + 1)));
    std::vector<unsigned long long> base((size_t)k + 1);
    for (uint32_t it = 0; it < d->iters; ++it) {
      ST_TRY(train_assign(w, d, n_rows));
      HIP_TRY(hipMemcpyAsync(hist.data(), w.hist.p, sizeof(uint32_t) * k, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      unsigned long long run = 0;
      for (uint32_t p = 0; p < k; ++p) {
        base[p] = run;
        run += hist[p];
      }
      base[k] = run;
      HIP_TRY(hipMemcpyAsync(w.po.p, base.data(), sizeof(unsigned long long) * ((size_t)k + 1), hipMemcpyHostToDevice, st));
      ST_TRY(stable_order(st, w.assign.as<uint32_t>(), n_rows, k, base, w.cntB, w.lrank, w.run, w.order.as<uint64_t>()));
      hipLaunchKernelGGL(k_centroid_update, dim3(k, (dim + 255) / 256), dim3(256), 0, st, w.xp.as<float>(),
                         w.order.as<uint64_t>(), w.po.as<unsigned long long>(), dim, w.cen.as<float>());
      HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(centroids, w.cen.p, sizeof(float) * (size_t)k * dim,
                           host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
  }
  if (out_counts) {
    std::vector<uint64_t> c64(hist.begin(), hist.end());
    HIP_TRY(hipStreamSynchronize(st));
    if (host)
      memcpy(out_counts, c64.data(), sizeof(uint64_t) * k);
    else
      HIP_TRY(hipMemcpy(out_counts, c64.data(), sizeof(uint64_t) * k, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipStreamSynchronize(st));
  return MI355_OK;
}

extern "C" int32_t mi355_ivf_residuals(const mi355_kmeans_desc* d, const float* vectors, uint64_t n_rows,
                                       const float* centroids, float* out_residuals, uint32_t* out_assign) {
  ST_TRY(check_kmeans_desc(d));
  if (!centroids || (n_rows && (!vectors || !out_residuals))) return fail(MI355_ERR_INVALID_INPUT, "NULL buffer");
  if (n_rows >> 40) return fail(MI355_ERR_NOT_SUPPORTED, "too many rows");
  ST_TRY(need_device(d->device));
  if (n_rows == 0) return MI355_OK;
  const uint32_t dim = d->dim, k = d->k;
  const bool host = d->mem == MI355_MEM_HOST;
  TrainScratch w;
  HIP_TRY(hipStreamCreateWithFlags(&w.st, hipStreamNonBlocking));
  hipStream_t st = w.st;
  ST_TRY(w.cen.ensure(sizeof(float) * (size_t)k * dim));
  HIP_TRY(copy_in(w.cen.p, centroids, sizeof(float) * (size_t)k * dim, d->mem, st));
  ST_TRY(train_load_rows(w, d, vectors, n_rows));
  ST_TRY(train_assign(w, d, n_rows));
  float* d_out = out_residuals;
  if (host) {
    ST_TRY(w.out.ensure(sizeof(float) * n_rows * dim));
    d_out = w.out.as<float>();
  }
  const uint64_t items = n_rows * dim, blocks = (items + 255) / 256;
  if (blocks > 0x7FFFFFFFull) return fail(MI355_ERR_NOT_SUPPORTED, "too many rows for one launch");
  hipLaunchKernelGGL(k_residuals, dim3((uint32_t)blocks), dim3(256), 0, st, w.xp.as<float>(), w.assign.as<uint32_t>(),
                     w.cen.as<float>(), n_rows, dim, d->metric, d_out);
  HIP_TRY(hipGetLastError());
  if (host) HIP_TRY(hipMemcpyAsync(out_residuals, d_out, sizeof(float) * n_rows * dim, hipMemcpyDeviceToHost, st));
  if (out_assign)
    HIP_TRY(hipMemcpyAsync(out_assign, w.assign.p, sizeof(uint32_t) * n_rows,
                           host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));
  return MI355_OK;
}

// -------------------------------------------------------------------- flat --
extern "C" int32_t mi355_flat_open(const mi355_flat_desc* d, mi355_flat** out) {
  if (!out) return fail(MI355_ERR_INVALID_INPUT, "out is NULL");
  *out = nullptr;
  if (!d) return fail(MI355_ERR_INVALID_INPUT, "desc is NULL");
  if (d->struct_size != sizeof(mi355_flat_desc))
    return fail(MI355_ERR_INVALID_INPUT, "mi355_flat_desc.struct_size %u != %zu (ABI mismatch)",
                d->struct_size, sizeof(mi355_flat_desc));
  if (d->dim == 0) return fail(MI355_ERR_INVALID_INPUT, "dim must be > 0");
  if (d->dtype > MI355_DTYPE_F16 || d->mem > MI355_MEM_DEVICE)
    return fail(MI355_ERR_INVALID_INPUT, "bad dtype / mem enum");
  if (d->n_rows && !d->vectors) return fail(MI355_ERR_INVALID_INPUT, "vectors is NULL");
  if (d->n_rows >= 0xFFFFFFF0ull) return fail(MI355_ERR_NOT_SUPPORTED, "flat column limited to 2^32-16 rows");
  if ((size_t)d->dim * 4 > 60u * 1024) return fail(MI355_ERR_NOT_SUPPORTED, "dim %u too large", d->dim);
  ST_TRY(need_device(d->device));
  mi355_flat* f = new (std::nothrow) mi355_flat();
  if (!f) return fail(MI355_ERR_RUNTIME, "out of host memory");
  f->device = d->device;
  f->dim = d->dim;
  f->dtype = d->dtype;
  f->n_rows = d->n_rows;
  auto bail = [&](int32_t s) {
    mi355_flat_close(f);
    return s;
  };
  if (hipStreamCreateWithFlags(&f->own_stream, hipStreamNonBlocking) != hipSuccess)
    return bail(fail(MI355_ERR_RUNTIME, "hipStreamCreate failed"));
  f->stream = f->own_stream;
  size_t vb = dtype_size(d->dtype) * (size_t)d->dim * d->n_rows;
  int32_t s = f->vectors.ensure(std::max<size_t>(vb, 16));
  if (s) return bail(s);
  if (copy_in(f->vectors.p, d->vectors, vb, d->mem, f->stream) != hipSuccess)
    return bail(fail(MI355_ERR_RUNTIME, "upload of the vector column failed"));
  if (d->row_ids) {
    s = f->row_ids.ensure(std::max<size_t>(sizeof(uint64_t) * d->n_rows, 16));
    if (s) return bail(s);
    if (copy_in(f->row_ids.p, d->row_ids, sizeof(uint64_t) * d->n_rows, d->mem, f->stream) != hipSuccess)
      return bail(fail(MI355_ERR_RUNTIME, "upload of row ids failed"));
    f->has_row_ids = true;
  }
  // MFMA filter data: bf16 shadow (if needed), per-row |v|^2 and its maximum
  {
    const char* fm = getenv("MI355_FLAT");  // dev knob: "exact" keeps the scalar sweep only
    const bool force_exact = fm && !strcmp(fm, "exact");
    if (!force_exact && d->n_rows >= env_u32("MI355_FLAT_MFMA_MIN_ROWS", 4096)) {
      f->dimp = (d->dim + 63u) & ~63u;
      f->shadowed = d->dtype != MI355_DTYPE_BF16 || f->dimp != d->dim;
      if (f->shadowed) {
        s = f->shadow.ensure((size_t)d->n_rows * f->dimp * 2);
        if (s) return bail(s);
      }
      // padded to whole 256-row tiles (tail = 0): the GEMM epilogue loads its tile's terms unconditionally
      const size_t vv_rows = (d->n_rows + 255) / 256 * 256;
      s = f->vv.ensure(sizeof(float) * vv_rows);
      if (s) return bail(s);
      if (hipMemsetAsync(f->vv.as<float>() + d->n_rows, 0, sizeof(float) * (vv_rows - d->n_rows), f->stream) != hipSuccess)
        return bail(fail(MI355_ERR_RUNTIME, "memset failed"));
      s = f->vmax.ensure(64);
      if (s) return bail(s);
      if (hipMemsetAsync(f->vmax.p, 0, 64, f->stream) != hipSuccess) return bail(fail(MI355_ERR_RUNTIME, "memset failed"));
      FlatRowPrepArgs ra;
      ra.vectors = f->vectors.p;
      ra.dtype = d->dtype;
      ra.dim = d->dim;
      ra.dimp = f->dimp;
      ra.n_rows = d->n_rows;
      ra.shadow = f->shadowed ? f->shadow.as<uint16_t>() : nullptr;
      ra.vv = f->vv.as<float>();
      ra.max_key = f->vmax.as<uint32_t>();
      hipLaunchKernelGGL(k_flat_prep_rows, dim3((uint32_t)((d->n_rows + 3) / 4)), dim3(256), 0, f->stream, ra);
      uint32_t key = 0;
      if (hipGetLastError() != hipSuccess ||
          hipMemcpyAsync(&key, f->vmax.p, 4, hipMemcpyDeviceToHost, f->stream) != hipSuccess ||
          hipStreamSynchronize(f->stream) != hipSuccess)
        return bail(fail(MI355_ERR_RUNTIME, "building the flat filter data failed"));
      // inverse of f32_sort_key for non-negative values
      uint32_t u = (key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key;
      memcpy(&f->vv_max, &u, 4);
      if (key == 0) f->vv_max = 0.f;
      // relative error of the bf16 dot product the filter must absorb: query rounding 2^-9,
      // row rounding (value + its norm) 3 * 2^-9 when the column was converted, accumulation
      f->c_err = ldexpf(1.f, -9) * (d->dtype != MI355_DTYPE_BF16 ? 4.f : 1.f) + (float)f->dimp * ldexpf(1.f, -22);
      f->mfma = true;
    }
  }
  if (hipStreamSynchronize(f->stream) != hipSuccess) return bail(fail(MI355_ERR_RUNTIME, "sync failed"));
  *out = f;
  return MI355_OK;
}

// The MFMA filter + exact re-rank over queries [d_q, d_q + n) (device), results in d_ids/d_dist/d_cnt
static int32_t run_flat_mfma(mi355_flat* f, const float* d_q, uint32_t nq, uint32_t metric, uint32_t k,
                             const RangeFilter& range, uint64_t* d_ids, float* d_dist, uint32_t* d_cnt) {
  hipStream_t st = f->stream;
  // tile shape: 256 x 256 (8 waves) for real batches, 128 x 128 (4 waves, 2 workgroups per CU) for small ones
  const uint32_t tile = nq > 128 ? env_u32("MI355_FLAT_TILE", 256) : 128;  // dev knob: 128, 256, 3 (= 256 x 128, 3 stages), 8 (experimental 8-phase)
  const bool oct = tile == 8;  // dev: EXPERIMENTAL 8-phase schedule (kernels_flat_mfma8.h), not validated on hardware
  const bool big = tile == 256 || oct, tri = tile == 3;
  const uint32_t BM = (big || tri) ? 256 : 128, BN = big ? 256 : 128;
  const uint32_t n_rtiles = (uint32_t)((f->n_rows + BM - 1) / BM);
  const uint32_t n_groups = n_rtiles * (BM / FG_GROUP);
  uint32_t groups_per_seg = (n_groups + FG_MAX_SEG - 1) / FG_MAX_SEG;
  const uint32_t n_seg = (n_groups + groups_per_seg - 1) / groups_per_seg;
  // bound the group-minimum matrix (n_groups x queries f32) to ~2 GiB per pass
  const size_t budget = (size_t)env_u32("MI355_WORKSPACE_MB", 2048) << 20;
  uint32_t chunk = (uint32_t)std::min<size_t>(((size_t)nq + BN - 1) / BN * BN,
                                              std::max<size_t>(BN, (budget / ((size_t)n_groups * 4)) / BN * BN));
  const int kpl = kpl_for(k);
  const bool dbg_sync = env_u32("MI355_FLAT_SYNC", 0) != 0;  // dev: synchronise after every stage to localise a fault
  auto stage_ok = [&](const char* what) -> int32_t {
    if (!dbg_sync) return MI355_OK;
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(MI355_ERR_RUNTIME, "flat stage %s failed: %s", what, hipGetErrorString(e));
    fprintf(stderr, "[mi355] flat stage %s ok\n", what);
    return MI355_OK;
  };
  ST_TRY(f->g_qb.ensure((size_t)chunk * f->dimp * 2));
  ST_TRY(f->g_qa.ensure(sizeof(float) * chunk));
  ST_TRY(f->g_qg.ensure(sizeof(float) * chunk));
  ST_TRY(f->g_slack.ensure(sizeof(float) * chunk));
  ST_TRY(f->g_tau.ensure(sizeof(float) * chunk));
  ST_TRY(f->g_gm.ensure(sizeof(float) * (size_t)n_groups * chunk));
  ST_TRY(f->g_seg.ensure(sizeof(float) * (size_t)n_seg * chunk));
  ST_TRY(f->g_cnt.ensure(sizeof(uint32_t) * chunk));
  ST_TRY(f->g_cand.ensure(sizeof(uint32_t) * (size_t)chunk * FG_CAND_CAP));
  for (uint32_t q0 = 0; q0 < nq; q0 += chunk) {
    const uint32_t n = std::min(chunk, nq - q0);
    const uint32_t n_pad = (n + BN - 1) / BN * BN;
    FlatQueryPrepArgs qa;
    qa.q = d_q + (size_t)q0 * f->dim;
    qa.nq = n;
    qa.nq_pad = n_pad;
    qa.dim = f->dim;
    qa.dimp = f->dimp;
    qa.metric = metric;
    qa.c_err = f->c_err;
    qa.vv_max = f->vv_max;
    qa.qb = f->g_qb.as<uint16_t>();
    qa.qa = f->g_qa.as<float>();
    qa.qg = f->g_qg.as<float>();
    qa.qslack = f->g_slack.as<float>();
    hipLaunchKernelGGL(k_flat_prep_queries, dim3((n_pad + 3) / 4), dim3(256), 0, st, qa);
    ST_TRY(stage_ok("prep_queries"));
    FlatGemmArgs ga;
    ga.v = f->shadowed ? f->shadow.as<uint16_t>() : (const uint16_t*)f->vectors.p;
    ga.qb = qa.qb;
    ga.vv = f->vv.as<float>();
    ga.qa = qa.qa;
    ga.qg = qa.qg;
    ga.n_rows = f->n_rows;
    ga.dimp = f->dimp;
    ga.nq_pad = n_pad;
    ga.n_qtiles = n_pad / BN;
    ga.n_rtiles = n_rtiles;
    ga.omc = 1.f - f->c_err;
    ga.gm = f->g_gm.as<float>();
    uint32_t gemm_blocks = ((n_rtiles + 7) / 8) * 8 * ga.n_qtiles;  // one per (row tile, query tile)
    // persistent grid: one workgroup per CU slot walks its XCD's tiles and overlaps the next tile's
    // first stage with the current tile's last k-step and epilogue.  MI355_FLAT_PERSIST: 0 = one
    // workgroup per tile, 1 = default, N >= 8 = a grid of N workgroups (dev / tests: forces the
    // cross-tile path on small columns)
    if (const uint32_t persist = oct ? 0u : env_u32("MI355_FLAT_PERSIST", 1)) {
      uint32_t slots = persist / 8 * 8;
      if (persist < 8) {
        int cus = 0;
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, f->device));
        slots = (uint32_t)std::max(cus, 8) / 8 * 8 * ((big || tri) ? 1u : 2u);
      }
      gemm_blocks = std::min(gemm_blocks, slots);
    }
    const size_t gemm_lds = (size_t)(tri ? 3 : 2) * (BM + BN) * FG_BK * 2;
#define LAUNCH_FG(MET)                                                                              \
  {                                                                                                 \
    if (tri) {                                                                                      \
      auto kern = k_flat_gemm<MET, 4, 2, 4, 4, 3>;                                                  \
      HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                  (int)gemm_lds));                                                  \
      hipLaunchKernelGGL(kern, dim3(gemm_blocks), dim3(512), gemm_lds, st, ga);                     \
    } else if (oct) {                                                                               \
      auto kern = k_flat_gemm8<MET>;                                                                \
      HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                  (int)gemm_lds));                                                  \
      hipLaunchKernelGGL(kern, dim3(gemm_blocks), dim3(512), gemm_lds, st, ga);                     \
    } else if (big) {                                                                               \
      auto kern = k_flat_gemm<MET, 2, 4, 8, 4, 2>;                                                  \
      HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                  (int)gemm_lds));                                                  \
      hipLaunchKernelGGL(kern, dim3(gemm_blocks), dim3(512), gemm_lds, st, ga);                     \
    } else {                                                                                        \
      auto kern = k_flat_gemm<MET, 2, 2, 4, 4, 2>;                                                     \
      HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                  (int)gemm_lds));                                                  \
      hipLaunchKernelGGL(kern, dim3(gemm_blocks), dim3(256), gemm_lds, st, ga);                     \
    }                                                                                               \
  }
    if (metric == MI355_METRIC_L2) LAUNCH_FG(MI355_METRIC_L2)
    else if (metric == MI355_METRIC_COSINE) LAUNCH_FG(MI355_METRIC_COSINE)
    else LAUNCH_FG(MI355_METRIC_DOT)
#undef LAUNCH_FG
    HIP_TRY(hipGetLastError());
    ST_TRY(stage_ok("gemm"));
    if (dbg_sync) {
      // only whole tiles' groups of real rows are defined; padding queries are computed too
      unsigned long long* d_sum = nullptr;
      unsigned long long h_sum = 0;
      HIP_TRY(hipMalloc(&d_sum, 8));
      HIP_TRY(hipMemsetAsync(d_sum, 0, 8, st));
      const size_t real_groups = (size_t)((f->n_rows + FG_GROUP - 1) / FG_GROUP);
      hipLaunchKernelGGL(k_flat_checksum, dim3(1024), dim3(256), 0, st, ga.gm, real_groups * n_pad, d_sum);
      HIP_TRY(hipMemcpyAsync(&h_sum, d_sum, 8, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      (void)hipFree(d_sum);
      fprintf(stderr, "[mi355] flat gm checksum %016llx (tile %u)\n", h_sum, tile);
    }
    hipLaunchKernelGGL(k_flat_segmin, dim3((n_pad + 255) / 256, n_seg), dim3(256), 0, st, ga.gm, n_groups, n_pad,
                       groups_per_seg, f->g_seg.as<float>());
    if (kpl == 1)
      hipLaunchKernelGGL(k_flat_tau<1>, dim3(n), dim3(64), 0, st, f->g_seg.as<float>(), n_seg, n_pad, k, qa.qslack, f->g_tau.as<float>(), f->g_cnt.as<uint32_t>());
    else if (kpl == 2)
      hipLaunchKernelGGL(k_flat_tau<2>, dim3(n), dim3(64), 0, st, f->g_seg.as<float>(), n_seg, n_pad, k, qa.qslack, f->g_tau.as<float>(), f->g_cnt.as<uint32_t>());
    else
      hipLaunchKernelGGL(k_flat_tau<4>, dim3(n), dim3(64), 0, st, f->g_seg.as<float>(), n_seg, n_pad, k, qa.qslack, f->g_tau.as<float>(), f->g_cnt.as<uint32_t>());
    ST_TRY(stage_ok("segmin+tau"));
    const uint32_t ysplit = std::min<uint32_t>(256, std::max<uint32_t>(1, n_groups / 512));
    hipLaunchKernelGGL(k_flat_compact, dim3((n + 255) / 256, ysplit), dim3(256), 0, st, ga.gm, n_groups, n_pad, n,
                       f->g_tau.as<float>(), f->g_cnt.as<uint32_t>(), f->g_cand.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    ST_TRY(stage_ok("compact"));
    FlatRerankArgs ra;
    ra.f.vectors = f->vectors.p;
    ra.f.dtype = f->dtype;
    ra.f.row_ids = f->has_row_ids ? f->row_ids.as<uint64_t>() : nullptr;
    ra.f.n_rows = f->n_rows;
    ra.f.dim = f->dim;
    ra.f.metric = metric;
    ra.f.q = d_q + (size_t)q0 * f->dim;
    ra.f.slice_rows = 0;
    ra.f.n_slices = 1;
    ra.f.kk = k;
    ra.f.range = range;
    ra.f.filter.mode = MI355_FILTER_NONE;
    ra.f.filter.pad = 0;
    ra.f.filter.ids = nullptr;
    ra.f.filter.n = 0;
    ra.f.cand = nullptr;
    ra.cand_cnt = f->g_cnt.as<uint32_t>();
    ra.cand = f->g_cand.as<uint32_t>();
    ra.out_ids = d_ids + (size_t)q0 * k;
    ra.out_dist = d_dist + (size_t)q0 * k;
    ra.out_cnt = d_cnt + q0;
    const size_t rl = (((size_t)f->dim * 4 + 15) & ~(size_t)15) + sizeof(Cand) * 4 * k;
    launch_by_kpl(kpl, k_flat_rerank<1>, k_flat_rerank<2>, k_flat_rerank<4>, dim3(n), dim3(256), rl, st, ra);
    HIP_TRY(hipGetLastError());
    ST_TRY(stage_ok("rerank"));
  }
  return MI355_OK;
}

extern "C" int32_t mi355_flat_close(mi355_flat* f) {
  if (!f) return MI355_OK;
  (void)hipSetDevice(f->device);
  DevBuf* bufs[] = {&f->vectors, &f->row_ids, &f->w_q,  &f->w_cand, &f->w_ids,  &f->w_dist, &f->w_cnt,
                    &f->shadow,  &f->vv,      &f->vmax, &f->g_qb,   &f->g_qa,   &f->g_qg,   &f->g_slack,
                    &f->g_tau,   &f->g_gm,    &f->g_seg, &f->g_cnt, &f->g_cand, &f->w_filter};
  for (DevBuf* b : bufs) b->release();
  if (f->own_stream) (void)hipStreamDestroy(f->own_stream);
  delete f;
  return MI355_OK;
}

extern "C" int32_t mi355_flat_set_stream(mi355_flat* f, void* hip_stream) {
  if (!f) return fail(MI355_ERR_INVALID_INPUT, "flat handle is NULL");
  std::lock_guard<std::mutex> lk(f->mu);
  f->stream = hip_stream ? (hipStream_t)hip_stream : f->own_stream;
  return MI355_OK;
}

extern "C" int32_t mi355_flat_sync(mi355_flat* f) {
  if (!f) return fail(MI355_ERR_INVALID_INPUT, "flat handle is NULL");
  HIP_TRY(hipSetDevice(f->device));
  HIP_TRY(hipStreamSynchronize(f->stream));
  return MI355_OK;
}

extern "C" int32_t mi355_flat_search(mi355_flat* f, const float* queries, uint32_t n_queries,
                                     const mi355_search_params* p, uint64_t* out_rowids,
                                     float* out_dist, uint32_t* out_counts) {
  if (!f) return fail(MI355_ERR_INVALID_INPUT, "flat handle is NULL");
  ST_TRY(validate_params(p));
  uint32_t metric = p->metric == MI355_METRIC_DEFAULT ? (uint32_t)MI355_METRIC_L2 : p->metric;
  if (metric > MI355_METRIC_DOT) return fail(MI355_ERR_INVALID_INPUT, "unknown metric %u", metric);
  if (n_queries == 0) return MI355_OK;
  if (!queries || !out_counts || (p->k && (!out_rowids || !out_dist)))
    return fail(MI355_ERR_INVALID_INPUT, "NULL query / output buffer");
  const uint32_t k = p->k;
  std::lock_guard<std::mutex> lk(f->mu);
  HIP_TRY(hipSetDevice(f->device));
  hipStream_t st = f->stream;
  const bool host_io = p->io_mem == MI355_MEM_HOST;
  if (k == 0) {
    if (host_io) memset(out_counts, 0, sizeof(uint32_t) * n_queries);
    else HIP_TRY(hipMemsetAsync(out_counts, 0, sizeof(uint32_t) * n_queries, st));
    return MI355_OK;
  }
  const int kpl = kpl_for(k);
  if (!kpl) return fail(MI355_ERR_NOT_SUPPORTED, "k = %u exceeds the supported 256", k);
  auto t_start = std::chrono::steady_clock::now();
  const float* d_q = queries;
  uint64_t* d_ids = out_rowids;
  float* d_dist = out_dist;
  uint32_t* d_cnt = out_counts;
  if (host_io) {
    ST_TRY(f->w_q.ensure(sizeof(float) * (size_t)n_queries * f->dim));
    ST_TRY(f->w_ids.ensure(sizeof(uint64_t) * (size_t)n_queries * k));
    ST_TRY(f->w_dist.ensure(sizeof(float) * (size_t)n_queries * k));
    ST_TRY(f->w_cnt.ensure(sizeof(uint32_t) * n_queries));
    HIP_TRY(hipMemcpyAsync(f->w_q.p, queries, sizeof(float) * (size_t)n_queries * f->dim, hipMemcpyHostToDevice, st));
    d_q = f->w_q.as<float>();
    d_ids = f->w_ids.as<uint64_t>();
    d_dist = f->w_dist.as<float>();
    d_cnt = f->w_cnt.as<uint32_t>();
  }
  RangeFilter rng;
  rng.has_lower = p->has_lower_bound;
  rng.has_upper = p->has_upper_bound;
  rng.lower = p->lower_bound;
  rng.upper = p->upper_bound;
  // MFMA filter + exact re-rank whenever the column carries the filter data.  A lower
  // bound makes "the k best" and "the k best in range" different sets: exact sweep.
  RowFilter flt;
  ST_TRY(make_row_filter(p, f->w_filter, st, &flt));
  // (the filter's k-th-best bound assumes every row is eligible: prefiltered searches sweep exactly)
  const bool use_mfma = f->mfma && !p->has_lower_bound && flt.mode == MI355_FILTER_NONE;
  f->last_path = use_mfma ? 1 : 2;
  if (use_mfma) {
    ST_TRY(run_flat_mfma(f, d_q, n_queries, metric, k, rng, d_ids, d_dist, d_cnt));
  } else {
  // enough work items to fill 256 CUs, at least 1024 rows each
  uint32_t slice = (uint32_t)std::max<uint64_t>(1024, (f->n_rows + 2047) / 2048);
  slice = (slice + 255u) & ~255u;
  const uint32_t n_slices = (uint32_t)std::max<uint64_t>(1, (f->n_rows + slice - 1) / slice);
  const uint32_t chunk = std::min(n_queries, 65535u);
  ST_TRY(f->w_cand.ensure(sizeof(Cand) * (size_t)chunk * n_slices * k));
  for (uint32_t q0 = 0; q0 < n_queries; q0 += chunk) {
    const uint32_t n = std::min(chunk, n_queries - q0);
    FlatArgs fa;
    fa.vectors = f->vectors.p;
    fa.dtype = f->dtype;
    fa.row_ids = f->has_row_ids ? f->row_ids.as<uint64_t>() : nullptr;
    fa.n_rows = f->n_rows;
    fa.dim = f->dim;
    fa.metric = metric;
    fa.q = d_q + (size_t)q0 * f->dim;
    fa.slice_rows = slice;
    fa.n_slices = n_slices;
    fa.kk = k;
    fa.range.has_lower = p->has_lower_bound;
    fa.range.has_upper = p->has_upper_bound;
    fa.range.lower = p->lower_bound;
    fa.range.upper = p->upper_bound;
    fa.filter = flt;
    fa.cand = f->w_cand.as<Cand>();
    size_t lds = (((size_t)f->dim * 4 + 15) & ~(size_t)15) + sizeof(Cand) * 4 * k;
    launch_by_kpl(kpl, k_flat_scan<1>, k_flat_scan<2>, k_flat_scan<4>, dim3(n_slices, 1, n), dim3(256), lds, st, fa);
    HIP_TRY(hipGetLastError());
    MergeArgs ma;
    ma.cand = f->w_cand.as<Cand>();
    ma.n_src = n_slices;
    ma.kk_in = k;
    ma.k_out = k;
    ma.out_ids = d_ids + (size_t)q0 * k;
    ma.out_dist = d_dist + (size_t)q0 * k;
    ma.out_pos = nullptr;
    ma.out_cnt = d_cnt + q0;
    launch_by_kpl(kpl, k_merge_cands<1>, k_merge_cands<2>, k_merge_cands<4>, dim3(n), dim3(64), 0, st, ma);
    HIP_TRY(hipGetLastError());
  }
  }  // exact sweep
  if (host_io) {
    HIP_TRY(hipMemcpyAsync(out_rowids, d_ids, sizeof(uint64_t) * (size_t)n_queries * k, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_dist, d_dist, sizeof(float) * (size_t)n_queries * k, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_counts, d_cnt, sizeof(uint32_t) * n_queries, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (p->timeout_ms) {
      auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_start).count();
      if (ms > (long long)p->timeout_ms)
        return fail(MI355_ERR_TIMEOUT, "Query timeout: %lld ms > %u ms", (long long)ms, p->timeout_ms);
    }
  }
  return MI355_OK;
}

extern "C" int32_t mi355_flat_info(const mi355_flat* f, uint32_t* out_last_path, uint32_t* out_has_filter) {
  if (!f) return fail(MI355_ERR_INVALID_INPUT, "flat handle is NULL");
  if (out_last_path) *out_last_path = f->last_path;
  if (out_has_filter) *out_has_filter = f->mfma ? 1u : 0u;
  return MI355_OK;
}

// ------------------------------------------------------------------- merge --
extern "C" int32_t mi355_merge_topk(int32_t device, void* hip_stream, const uint64_t* in_rowids,
                                    const float* in_dist, const uint32_t* in_counts,
                                    uint32_t n_lists, uint32_t n_queries, uint32_t k,
                                    uint64_t* out_rowids, float* out_dist, uint32_t* out_counts) {
  if (n_queries == 0) return MI355_OK;
  if (!in_rowids || !in_dist || !in_counts || !out_rowids || !out_dist || !out_counts)
    return fail(MI355_ERR_INVALID_INPUT, "NULL buffer");
  if (n_lists == 0 || k == 0) return fail(MI355_ERR_INVALID_INPUT, "n_lists and k must be > 0");
  const int kpl = kpl_for(k);
  if (!kpl) return fail(MI355_ERR_NOT_SUPPORTED, "k = %u exceeds the supported 256", k);
  ST_TRY(need_device(device));
  hipStream_t st = (hipStream_t)hip_stream;
  if (kpl == 1)
    hipLaunchKernelGGL(k_merge_lists<1>, dim3(n_queries), dim3(64), 0, st, in_rowids, in_dist, in_counts, n_lists, n_queries, k, out_rowids, out_dist, out_counts);
  else if (kpl == 2)
    hipLaunchKernelGGL(k_merge_lists<2>, dim3(n_queries), dim3(64), 0, st, in_rowids, in_dist, in_counts, n_lists, n_queries, k, out_rowids, out_dist, out_counts);
  else
    hipLaunchKernelGGL(k_merge_lists<4>, dim3(n_queries), dim3(64), 0, st, in_rowids, in_dist, in_counts, n_lists, n_queries, k, out_rowids, out_dist, out_counts);
  HIP_TRY(hipGetLastError());
  return MI355_OK;
}
