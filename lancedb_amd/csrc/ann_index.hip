// ann_index.hip — the IVF-PQ handle behind include/mi355_ann.h: host-side planner +
// launches of the gfx950 kernels.
//
// Replaces, behind lancedb::query::VectorQuery, what
// /root/reference/rust/lancedb/src/table/query.rs:219-327 hands to the lance
// Scanner (E1 in SURVEY.md §2b): nearest / nprobes / refine / distance_range /
// use_index -> a fixed launch sequence per query batch instead of a DataFusion
// plan.
#include "ann_internal.h"
#include "kernels_ivfpq.h"
#include "kernels_skew.h"



// Page-locked caller ranges (MI355_INDEX_RAW_HOST_MAPPED), reference counted per process: several
// handles (e.g. the shard handles of one column) may map the same range, which must stay registered
// until the last of them closes.
namespace {
struct HostMap {
  size_t bytes;
  uint32_t refs;
  void* dev;
};
std::mutex g_hostmap_mu;
// never destroyed: handles may be closed by the host's finalisers after this library's static
// destructors have run (e.g. a Python interpreter shutting down)
std::map<void*, HostMap>& g_hostmap = *new std::map<void*, HostMap>();
}  // namespace

static int32_t hostmap_acquire(void* host, size_t bytes, const void** out_dev) {
  std::lock_guard<std::mutex> lk(g_hostmap_mu);
  auto it = g_hostmap.find(host);
  if (it != g_hostmap.end()) {
    if (it->second.bytes < bytes)
      return fail(MI355_ERR_INVALID_INPUT, "host range %p is already mapped with %zu B, now %zu B are asked for", host,
                  it->second.bytes, bytes);
    ++it->second.refs;
    *out_dev = it->second.dev;
    return MI355_OK;
  }
  hipError_t e = hipHostRegister(host, bytes, hipHostRegisterMapped);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(MI355_ERR_RUNTIME, "hipHostRegister of %zu B of raw vectors failed: %s", bytes, hipGetErrorString(e));
  }
  void* dp = nullptr;
  e = hipHostGetDevicePointer(&dp, host, 0);
  if (e != hipSuccess) {
    (void)hipHostUnregister(host);
    (void)hipGetLastError();
    return fail(MI355_ERR_RUNTIME, "hipHostGetDevicePointer failed: %s", hipGetErrorString(e));
  }
  g_hostmap[host] = HostMap{bytes, 1u, dp};
  *out_dev = dp;
  return MI355_OK;
}

static void hostmap_release(void* host) {
  std::lock_guard<std::mutex> lk(g_hostmap_mu);
  auto it = g_hostmap.find(host);
  if (it == g_hostmap.end()) return;
  if (--it->second.refs == 0) {
    (void)hipHostUnregister(host);
    (void)hipGetLastError();  // never leave a sticky error behind for the next call's hipGetLastError()
    g_hostmap.erase(it);
  }
}

IndexView make_view(const mi355_index* ix) {
  IndexView v;
  v.dim = ix->dim;
  v.nlist = ix->nlist;
  v.m = ix->m;
  v.dsub = ix->dsub;
  v.metric = ix->metric;
  v.nbits = ix->nbits;
  v.mb = ix->mb;
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
  v.raw = ix->has_raw ? (ix->raw_mapped_dev ? ix->raw_mapped_dev : ix->raw.p) : nullptr;
  v.raw_dtype = ix->raw_dtype;
  v.raw_by_global = (ix->raw_mapped_dev && !ix->local_arrays) ? 1u : 0u;
  if (ix->raw_attached) {  // a borrowed device column in local row order takes precedence
    v.raw = ix->raw_attached;
    v.raw_dtype = ix->raw_attached_dtype;
    v.raw_by_global = 0;
  }
  return v;
}
// ------------------------------------------------------------- index open ---
static int32_t validate_index_desc(const mi355_index_desc* d) {
  if (!d) return fail(MI355_ERR_INVALID_INPUT, "desc is NULL");
  if (d->struct_size != sizeof(mi355_index_desc))
    return fail(MI355_ERR_INVALID_INPUT, "mi355_index_desc.struct_size %u != %zu (ABI mismatch)",
                d->struct_size, sizeof(mi355_index_desc));
  if (d->nbits != 8 && d->nbits != 4) return fail(MI355_ERR_INVALID_INPUT, "num_bits must be 4 or 8, got %u", d->nbits);
  if (d->dim == 0 || d->nlist == 0 || d->m == 0)
    return fail(MI355_ERR_INVALID_INPUT, "dim, nlist and m must be > 0");
  if (d->dim % d->m != 0)
    return fail(MI355_ERR_INVALID_INPUT, "dim %u is not divisible by num_sub_vectors %u", d->dim, d->m);
  // table/create_index.rs:96-101: 4-bit codes are packed two per byte
  if (d->nbits == 4 && d->m % 2 != 0)
    return fail(MI355_ERR_INVALID_INPUT, "num_sub_vectors must be even when num_bits is 4, got %u", d->m);
  if (d->flags & ~(uint32_t)(MI355_INDEX_GENERIC_SCAN | MI355_INDEX_RAW_HOST_MAPPED | MI355_INDEX_LOCAL_ARRAYS))
    return fail(MI355_ERR_INVALID_INPUT, "unknown index flags 0x%x", d->flags);
  if ((d->flags & MI355_INDEX_LOCAL_ARRAYS) && d->n_rows && !d->row_ids)
    return fail(MI355_ERR_INVALID_INPUT, "MI355_INDEX_LOCAL_ARRAYS needs row_ids (identity ids would be global positions)");
  if ((d->flags & MI355_INDEX_RAW_HOST_MAPPED) && (d->mem != MI355_MEM_HOST || !d->raw_vectors))
    return fail(MI355_ERR_INVALID_INPUT, "MI355_INDEX_RAW_HOST_MAPPED needs host raw_vectors (mem = MI355_MEM_HOST)");
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
  if (d->part_owner && d->shard_count > 1)
    for (uint32_t p = 0; p < d->nlist; ++p)
      if (d->part_owner[p] >= d->shard_count)
        return fail(MI355_ERR_INVALID_INPUT, "part_owner[%u] = %u is not a shard of %u", p, d->part_owner[p], d->shard_count);
  // an 8-bit distance table larger than the LDS keeps its tail in global memory (k_scan_pair SPILL);
  // what cannot work is a residual + candidate lists that leave no room for any table
  if (scan_pair_m_lds(d->m, d->nbits, d->dim) == 0)
    return fail(MI355_ERR_NOT_SUPPORTED, "dim %u / %u sub-vectors x %u entries do not fit the 160 KiB LDS", d->dim,
                d->m, 1u << d->nbits);
  return MI355_OK;
}

static int32_t index_free(mi355_index* ix) {
  if (!ix) return MI355_OK;
  (void)hipSetDevice(ix->device);
  DevBuf* bufs[] = {&ix->centroids, &ix->cnorm,  &ix->codebook, &ix->codes,   &ix->code_off,
                    &ix->plen,      &ix->pstride, &ix->lrow0,    &ix->grow0,   &ix->row_ids,
                    &ix->raw,       &ix->w_q,    &ix->w_qp,     &ix->w_qq,    &ix->w_coarse,
                    &ix->w_probes,  &ix->w_cand, &ix->w_ids,    &ix->w_dist,  &ix->w_pos,
                    &ix->w_cnt,     &ix->w_ids2, &ix->w_dist2,  &ix->w_cnt2,  &ix->w_ctl,
                    &ix->cbT,       &ix->order,  &ix->xcd_first, &ix->p_cnt,  &ix->p_off,
                    &ix->p_fill,    &ix->q_start, &ix->heads,   &ix->items,   &ix->qthr,
                    &ix->w_filter,  &ix->w_probes64, &ix->w_cand2, &ix->w_sq,     &ix->w_sids,
                    &ix->w_sdist,   &ix->w_scnt,     &ix->w_scnt_ann, &ix->w_spill, &ix->w_srows, &ix->w_ccnt,
                    &ix->w_partial,  &ix->w_cand2b,   &ix->w_cnt2b};
  for (DevBuf* b : bufs) b->release();
  for (auto* v : {&ix->ev_free, &ix->ev_pending})
    for (auto& es : *v)
      for (auto& e : es.ev) (void)hipEventDestroy(e);
  for (auto& kv : ix->graphs)
    if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
  if (ix->raw_mapped_host) hostmap_release(ix->raw_mapped_host);
  if (ix->xdone) (void)hipEventDestroy(ix->xdone);
  if (ix->h_pin) (void)hipHostFree(ix->h_pin);
  for (int i = 0; i < 2; ++i) {
    if (ix->r_scan[i]) (void)hipEventDestroy(ix->r_scan[i]);
    if (ix->r_done[i]) (void)hipEventDestroy(ix->r_done[i]);
  }
  if (ix->rstream) (void)hipStreamDestroy(ix->rstream);
  if (ix->own_stream) (void)hipStreamDestroy(ix->own_stream);
  (void)hipGetLastError();  // never leave a sticky error of the teardown behind for the thread's next launch check
  delete ix;
  return MI355_OK;
}

static int32_t index_open_impl(const mi355_index_desc* d, mi355_index* ix) {
  ix->device = d->device;
  ix->dim = d->dim;
  ix->nlist = d->nlist;
  ix->m = d->m;
  ix->dsub = d->dim / d->m;
  ix->nbits = d->nbits;
  ix->mb = d->m * d->nbits / 8;
  ix->metric = d->metric;
  ix->local_arrays = (d->flags & MI355_INDEX_LOCAL_ARRAYS) != 0;
  ix->shard_count = d->shard_count > 1 ? d->shard_count : 1;
  ix->shard_rank = d->shard_count > 1 ? d->shard_rank : 0;
  HIP_TRY(hipStreamCreateWithFlags(&ix->own_stream, hipStreamNonBlocking));
  ix->stream = ix->own_stream;
  hipStream_t st = ix->stream;
  const uint32_t nlist = d->nlist, m = d->m, mb = ix->mb, cb_entries = 1u << d->nbits;
  for (DevBuf* b : {&ix->w_q, &ix->w_qp, &ix->w_qq, &ix->w_coarse, &ix->w_probes, &ix->w_cand, &ix->w_ids, &ix->w_dist,
                    &ix->w_pos, &ix->w_cnt, &ix->w_ids2, &ix->w_dist2, &ix->w_cnt2, &ix->w_cand2, &ix->items, &ix->qthr, &ix->w_ccnt,
                    &ix->w_filter, &ix->w_probes64, &ix->w_spill, &ix->w_partial, &ix->w_cand2b, &ix->w_cnt2b})
    b->gen = &ix->ws_gen;  // a re-allocation of any of these invalidates the cached hipGraphs

  // -- ownership + local layout
  std::vector<uint32_t> owner;
  if (d->part_owner && ix->shard_count > 1)
    owner.assign(d->part_owner, d->part_owner + nlist);  // the caller's plan (validated: every id < shard_count)
  else
    shard_plan_host(d->part_offsets, nlist, ix->shard_count, owner);
  std::vector<uint32_t> plen(nlist), pstride(nlist), lrow0(nlist);
  std::vector<uint64_t> code_off(nlist), grow0(nlist);
  uint64_t rows = 0, bytes = 0;
  uint32_t owned = 0, max_len = 0;
  {
    const bool force_pair = (d->flags & MI355_INDEX_GENERIC_SCAN) != 0;
    // The production scan takes every 8-bit m (SkewShape: padded to a kernel width, or cut into slabs of <= 96
    // columns) whose work item fits the LDS: the 256 x 128-dword table + one slab's residual (the whole row's when
    // there is one slab) + the candidate lists of eight waves; a thread stages at most four residual elements.
    SkewShape shp{};
    ix->layout = MI355_SCAN_PAIR;
    if (!force_pair && d->nbits == 8 && sk_shape(m, &shp)) {
      const uint32_t res_floats = shp.n_slabs > 1 ? shp.M * ix->dsub : d->dim;
      if (res_floats <= 2048 && sk_scan_lds(res_floats, 8, 5) <= 160u * 1024) {
        ix->layout = MI355_SCAN_SKEW;
        ix->sk_M = shp.M;
        ix->sk_slabs = shp.n_slabs;
        ix->sk_slabbed = shp.slabbed;
        ix->sk_res_floats = res_floats;
      }
    }
  }
  const bool skew = ix->layout == MI355_SCAN_SKEW;
  const bool local_arrays = (d->flags & MI355_INDEX_LOCAL_ARRAYS) != 0;
  for (uint32_t p = 0; p < nlist; ++p) {
    uint64_t len = d->part_offsets[p + 1] - d->part_offsets[p];
    bool mine = owner[p] == ix->shard_rank;
    plen[p] = mine ? (uint32_t)len : 0;
    pstride[p] = (plen[p] + 15u) & ~15u;
    lrow0[p] = (uint32_t)rows;
    grow0[p] = d->part_offsets[p];
    code_off[p] = bytes;
    rows += plen[p];
    bytes += skew ? (uint64_t)ix->sk_slabs * sk_part_chunks((plen[p] + SK_TILE - 1) / SK_TILE, ix->sk_M / 16) * 1024u
                  : (uint64_t)mb * pstride[p];
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
  ST_TRY(ix->codebook.ensure(sizeof(float) * (size_t)m * cb_entries * ix->dsub));
  ST_TRY(ix->code_off.ensure(sizeof(uint64_t) * nlist));
  ST_TRY(ix->plen.ensure(sizeof(uint32_t) * nlist));
  ST_TRY(ix->pstride.ensure(sizeof(uint32_t) * nlist));
  ST_TRY(ix->lrow0.ensure(sizeof(uint32_t) * nlist));
  ST_TRY(ix->grow0.ensure(sizeof(uint64_t) * nlist));
  HIP_TRY(copy_in(ix->centroids.p, d->centroids, sizeof(float) * (size_t)nlist * d->dim, d->mem, st));
  HIP_TRY(copy_in(ix->codebook.p, d->codebook, sizeof(float) * (size_t)m * cb_entries * ix->dsub, d->mem, st));
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
    const size_t STAGE = (size_t)dev_knob("MI355_STAGE_MB", 256) << 20;
    ScratchBuf stage, d_srcoff, d_pids;
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
      ra.m = mb;  // code BYTES per row
      ra.transposed = d->codes_layout == MI355_CODES_PART_TRANSPOSED;
      SkewPackArgs sp;
      sp.src = ra.src;
      sp.src_off = ra.src_off;
      sp.part_ids = ra.part_ids;
      sp.dst = ra.dst;
      sp.code_off = ra.code_off;
      sp.plen = ra.plen;
      sp.m = ix->sk_M;
      sp.m_src = m;
      sp.transposed = ra.transposed;
      // grid.y is limited to 65535: split very wide batches
      for (size_t y0 = 0; y0 < pids.size(); y0 += 32768) {
        uint32_t ny = (uint32_t)std::min<size_t>(32768, pids.size() - y0);
        if (skew) {
          SkewPackArgs sb = sp;
          sb.src_off += y0;
          sb.part_ids += y0;
          hipLaunchKernelGGL(k_pack_skew, dim3(sk_pack_slots(batch_max_stride), ny, ix->sk_slabs), dim3(256),
                             2 * 64 * (ix->sk_M + 1), st, sb);
        } else {
          RepackArgs rb = ra;
          rb.src_off += y0;
          rb.part_ids += y0;
          hipLaunchKernelGGL(k_repack_codes, dim3((batch_max_stride + 63) / 64, ny), dim3(256),
                             64 * (mb + 1), st, rb);
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
      size_t pbytes = (size_t)mb * plen[p];
      uint64_t soff = (uint64_t)mb * (local_arrays ? (uint64_t)lrow0[p] : d->part_offsets[p]);
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
    order.resize(2 * (size_t)nlist);  // second half: the inverse permutation (k_plan_sparse)
    for (uint32_t at = 0; at < nlist; ++at) order[nlist + order[at]] = at;
    ST_TRY(ix->order.ensure(sizeof(uint32_t) * 2 * nlist));
    ST_TRY(ix->xcd_first.ensure(sizeof(uint32_t) * 9));
    ST_TRY(ix->p_cnt.ensure(sizeof(uint32_t) * 2 * nlist));  // two item classes per partition (PlanArgs::best_first)
    ST_TRY(ix->p_off.ensure(sizeof(uint32_t) * 2 * nlist));
    ST_TRY(ix->p_fill.ensure(sizeof(uint32_t) * 2 * nlist));
    ST_TRY(ix->q_start.ensure(sizeof(uint32_t) * 16));
    ST_TRY(ix->heads.ensure(sizeof(uint32_t) * 8 * SK_HEAD_STRIDE));
    HIP_TRY(hipMemcpyAsync(ix->order.p, order.data(), sizeof(uint32_t) * 2 * nlist, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ix->xcd_first.p, xcd_first.data(), sizeof(uint32_t) * 9, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(ix->p_cnt.p, 0, sizeof(uint32_t) * 2 * nlist, st));
    HIP_TRY(hipStreamSynchronize(st));  // host vectors above go out of scope
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ix->device));
    ix->n_cus = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
  }

  // -- row ids and raw vectors: owned partitions, concatenated in local order
  auto gather_rows = [&](DevBuf& dst, const void* src, size_t row_bytes) -> int32_t {
    ST_TRY(dst.ensure(std::max<size_t>(row_bytes * rows, 16)));
    if (local_arrays) {  // already this shard's rows in local order
      HIP_TRY(copy_in(dst.p, src, row_bytes * rows, d->mem, st));
      return MI355_OK;
    }
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
  if (d->raw_vectors && (d->flags & MI355_INDEX_RAW_HOST_MAPPED)) {
    // the column stays where it is (C5: 100 M x 1536 does not fit HBM): page-lock the caller's
    // range and let the refine kernel gather its k * refine_factor rows per query over PCIe.
    // Rows are addressed by GLOBAL index position (k_refine_dist converts local positions).
    const size_t bytes = dtype_size(d->raw_dtype) * (size_t)d->dim * (local_arrays ? rows : d->n_rows);
    if (bytes) {
      ST_TRY(hostmap_acquire(const_cast<void*>(d->raw_vectors), bytes, &ix->raw_mapped_dev));
      ix->raw_mapped_host = const_cast<void*>(d->raw_vectors);
    }
    ix->has_raw = true;
    ix->raw_dtype = d->raw_dtype;
    ix->raw_is_host = true;
  } else if (d->raw_vectors) {
    ST_TRY(gather_rows(ix->raw, d->raw_vectors, dtype_size(d->raw_dtype) * d->dim));
    ix->has_raw = true;
    ix->raw_dtype = d->raw_dtype;
  }
  ST_TRY(ix->w_ctl.ensure(sizeof(DevCtl)));
  HIP_TRY(hipMemsetAsync(ix->w_ctl.p, 0, sizeof(DevCtl), st));
  {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ix->device) == hipSuccess && khz > 0)
      ix->wall_khz = (uint32_t)khz;
    else
      (void)hipGetLastError();
  }
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
  ST_TRY(join_exchange(ix));
  ix->stream = hip_stream ? (hipStream_t)hip_stream : ix->own_stream;
  return MI355_OK;
}

extern "C" int32_t mi355_index_sync(mi355_index* ix) {
  if (!ix) return fail(MI355_ERR_INVALID_INPUT, "index is NULL");
  HIP_TRY(hipSetDevice(ix->device));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  if (ix->xdone) HIP_TRY(hipEventSynchronize(ix->xdone));  // an overlapped sharded search finishes on the communicator's stream
  // device-I/O calls cannot return their timeout: it is reported here (and in mi355_last_stats)
  uint32_t timed_out = 0;
  HIP_TRY(hipMemcpy(&timed_out, &ix->w_ctl.as<DevCtl>()->timed_out, 4, hipMemcpyDeviceToHost));
  if (timed_out) return fail(MI355_ERR_TIMEOUT, "Query timeout: the last search was stopped on the device");
  return MI355_OK;
}

extern "C" int32_t mi355_index_configure(mi355_index* ix, uint32_t scan_variant,
                                         uint32_t slice_rows, uint32_t profile) {
  if (!ix) return fail(MI355_ERR_INVALID_INPUT, "index is NULL");
  if (scan_variant > MI355_SCAN_SKEW) return fail(MI355_ERR_INVALID_INPUT, "unknown scan variant");
  if ((profile & MI355_PROFILE_MASK) > 2 ||
      (profile & ~(uint32_t)(MI355_PROFILE_MASK | MI355_CFG_GRAPH | MI355_CFG_COALESCE | MI355_CFG_DEFER_REFINE)))
    return fail(MI355_ERR_INVALID_INPUT, "unknown profile / mode bits 0x%x", profile);
  if (scan_variant != MI355_SCAN_AUTO && scan_variant != ix->layout)
    return fail(MI355_ERR_INVALID_INPUT,
                "scan variant %u does not match the code layout this index was packed for (%u)",
                scan_variant, ix->layout);
  std::lock_guard<std::mutex> lk(ix->mu);
  ix->scan_variant = scan_variant;
  ix->slice_rows = (slice_rows + 15u) & ~15u;
  ix->profile = profile & MI355_PROFILE_MASK;
  ix->use_graph = (profile & MI355_CFG_GRAPH) != 0;
  ix->coalesce = (profile & MI355_CFG_COALESCE) != 0;
  ix->defer_cfg = (profile & MI355_CFG_DEFER_REFINE) != 0;
  ++ix->ws_gen;  // captured graphs bake in the slicing
  HIP_TRY(hipSetDevice(ix->device));
  ST_TRY(join_exchange(ix));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  ST_TRY(drain_events(ix, true));
  reset_stats(ix);
  HIP_TRY(hipMemset(ix->w_ctl.p, 0, sizeof(DevCtl)));
  return MI355_OK;
}

extern "C" int32_t mi355_index_attach_raw(mi355_index* ix, const void* raw_vectors, uint32_t raw_dtype) {
  if (!ix || !raw_vectors) return fail(MI355_ERR_INVALID_INPUT, "NULL argument");
  if (raw_dtype > MI355_DTYPE_F16) return fail(MI355_ERR_INVALID_INPUT, "bad raw_dtype enum");
  std::lock_guard<std::mutex> lk(ix->mu);
  HIP_TRY(hipSetDevice(ix->device));
  ST_TRY(join_exchange(ix));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  ix->raw_attached = raw_vectors;
  ix->raw_attached_dtype = raw_dtype;
  {  // a borrowed column that lives in (mapped) HOST memory makes the re-rank a PCIe gather: it is then deferred
    hipPointerAttribute_t at{};
    ix->raw_is_host = hipPointerGetAttributes(&at, raw_vectors) == hipSuccess && at.type == hipMemoryTypeHost;
    (void)hipGetLastError();
  }
  ++ix->ws_gen;  // captured graphs hold the old column's address
  return MI355_OK;
}

extern "C" int32_t mi355_index_detach_raw(mi355_index* ix) {
  if (!ix) return fail(MI355_ERR_INVALID_INPUT, "index is NULL");
  std::lock_guard<std::mutex> lk(ix->mu);
  HIP_TRY(hipSetDevice(ix->device));
  ST_TRY(join_exchange(ix));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  ix->raw_attached = nullptr;
  ix->raw_is_host = ix->raw_mapped_dev != nullptr;
  ++ix->ws_gen;
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
int32_t drain_events(mi355_index* ix, bool discard) {
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

void reset_stats(mi355_index* ix) {
  ix->stats = mi355_stats{};
  ix->stats.struct_size = sizeof(mi355_stats);
}

extern "C" int32_t mi355_last_stats(mi355_index* ix, mi355_stats* out) {
  if (!ix || !out) return fail(MI355_ERR_INVALID_INPUT, "NULL argument");
  if (out->struct_size != sizeof(mi355_stats))
    return fail(MI355_ERR_INVALID_INPUT, "mi355_stats.struct_size mismatch");
  std::lock_guard<std::mutex> lk(ix->mu);
  HIP_TRY(hipSetDevice(ix->device));
  ST_TRY(join_exchange(ix));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  ST_TRY(drain_events(ix, false));
  DevCtl h_ctl;
  HIP_TRY(hipMemcpy(&h_ctl, ix->w_ctl.p, sizeof(DevCtl), hipMemcpyDeviceToHost));
  ix->stats.vectors_scanned = h_ctl.rows_scanned;
  ix->stats.code_bytes_scanned = h_ctl.rows_scanned * ix->mb;  // algorithmic bytes: m * nbits / 8 per vector
  ix->stats.timed_out = h_ctl.timed_out;
  ix->stats.bad_probes = h_ctl.bad_probes;
  *out = ix->stats;
  // queries re-searched over maximum_nprobes partitions were counted on the device
  out->n_queries += h_ctl.short_queries;
  out->partitions_probed += (uint64_t)h_ctl.short_queries * ix->second_np;
  out->struct_size = sizeof(mi355_stats);
  return MI355_OK;
}

#ifdef MI355_DEV_COUNTERS
// dev builds only (never in the product library): the scan's phase ticks and selection counters
extern "C" int32_t mi355_dev_counters(mi355_index* ix, uint32_t* out8, int32_t reset) {
  HIP_TRY(hipSetDevice(ix->device));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  DevCtl h;
  HIP_TRY(hipMemcpy(&h, ix->w_ctl.p, sizeof h, hipMemcpyDeviceToHost));
  memcpy(out8, h.dev, sizeof h.dev);
  if (reset) HIP_TRY(hipMemset(ix->w_ctl.as<DevCtl>()->dev, 0, sizeof h.dev));
  return MI355_OK;
}
#endif

// ------------------------------------------------------------------ search --
int32_t validate_params(const mi355_search_params* p) {
  if (!p) return fail(MI355_ERR_INVALID_INPUT, "params is NULL");
  if (p->struct_size != sizeof(mi355_search_params))
    return fail(MI355_ERR_INVALID_INPUT, "mi355_search_params.struct_size %u != %zu (ABI mismatch)",
                p->struct_size, sizeof(mi355_search_params));
  if (p->io_mem > MI355_MEM_DEVICE) return fail(MI355_ERR_INVALID_INPUT, "bad io_mem");
  if (p->filter_mode > MI355_FILTER_BLOCK) return fail(MI355_ERR_INVALID_INPUT, "unknown filter_mode %u", p->filter_mode);
  if (p->filter_mode != MI355_FILTER_NONE && p->n_filter && !p->filter_rowids)
    return fail(MI355_ERR_INVALID_INPUT, "filter_rowids is NULL");
  if (p->approx_mode > MI355_APPROX_ACCURATE)  // lib.rs:343-357
    return fail(MI355_ERR_INVALID_INPUT, "approx_mode must be one of 'fast', 'normal', or 'accurate', got %u", p->approx_mode);
  return MI355_OK;
}

// device view of the prefilter; a host array is staged into `stage`
int32_t make_row_filter(const mi355_search_params* p, DevBuf& stage, hipStream_t st, RowFilter* out) {
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


// exact distances of the ANN winners `in` [nq, kk] (only those owned by `my_rank` when `owner` is
// given) into `out` [nq, kk]; the top-k over them is a k_merge_cands launch by the caller
int32_t launch_refine(mi355_index* ix, const IndexView& view, const float* q, uint32_t nq, const Cand* in,
                      const uint32_t* in_cnt, const uint32_t* owner, uint32_t my_rank, uint32_t kk,
                      const RangeFilter& range, Cand* out, hipStream_t st, ActiveMask act, uint32_t max_blocks_y) {
  RefineArgs ra;
  ra.ix = view;
  ra.q = q;
  ra.in = in;
  ra.in_cnt = in_cnt;
  ra.in_owner = owner;
  ra.my_rank = my_rank;
  ra.kk = kk;
  ra.range = range;
  ra.out = out;
  ra.ctl = ix->w_ctl.as<DevCtl>();
  ra.n_rows = (uint32_t)ix->n_local;
  ra.act = act;
  const size_t rl = ((size_t)ix->dim * 4 + 15) & ~(size_t)15;
  for (uint32_t q0 = 0; q0 < nq; q0 += 65535u) {  // grid.y limit
    const uint32_t n = std::min(65535u, nq - q0);
    RefineArgs rb = ra;
    rb.q = q + (size_t)q0 * ix->dim;
    rb.in = in + (size_t)q0 * kk;
    rb.in_cnt = in_cnt + q0;
    rb.in_owner = owner ? owner + (size_t)q0 * kk : nullptr;
    rb.out = out + (size_t)q0 * kk;
    rb.act.base = act.base + q0;
    rb.nq = n;
    rb.side_slots = kk <= 64 ? 64u : kk <= 128 ? 128u : 256u;
    if (max_blocks_y)  // the re-rank beside the next call's scan: a few LDS-free workgroups striding over the queries
      hipLaunchKernelGGL(k_refine_dist<true>, dim3((kk + 255) / 256, std::min(n, max_blocks_y)), dim3(256), 0, st, rb);
    else
      hipLaunchKernelGGL(k_refine_dist<false>, dim3((kk + 255) / 256, n), dim3(256), rl, st, rb);
  }
  HIP_TRY(hipGetLastError());
  return MI355_OK;
}

// one pass of the pipeline over `nq` queries already resident at d_q;
// results land in d_ids/d_dist/d_cnt (device, [nq,k])
// d_cnt_ann [nq]: rows the ANN stage found per query, BEFORE the refine re-rank (what
// maximum_nprobes compares with k * refine_factor); may alias d_cnt when there is no refine
int32_t run_ivfpq(mi355_index* ix, const float* d_q, uint32_t nq, const SearchPlan& pl, uint64_t* d_ids,
                  float* d_dist, uint32_t* d_cnt, uint32_t* d_cnt_ann) {
  hipStream_t st = ix->stream;
  const IndexView view = make_view(ix);
  const uint32_t nprobe = pl.nprobe;
  const int kpl_kk = kpl_for(pl.kk), kpl_k = kpl_for(pl.k);

  const bool skew = ix->layout == MI355_SCAN_SKEW;
  // tuning (dev knobs; defaults chosen from the index shape)
  uint32_t nt = dev_knob("MI355_SCAN_THREADS", 0), vpt = dev_knob("MI355_SCAN_VPT", 0);
  if (!nt) nt = pl.kk > 64 ? 256 : ix->max_len >= 8192 ? 1024 : ix->max_len >= 2048 ? 512 : 256;
  // an 8-bit distance table larger than the LDS keeps its tail in global memory (long lists, 256 threads)
  const uint32_t m_lds = skew ? ix->m : scan_pair_m_lds(ix->m, ix->nbits, ix->dim);  // (the production scan never spills: it walks slabs)
  if (m_lds < ix->m) nt = 256;
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
  // the production scan slices by tile positions instead (SkewArgs::n_slices): only when the batch cannot
  // give every CU a work item, and never below ~2 k rows per slice (each slice rebuilds the distance table)
  uint32_t sk_slices = 1;
  if (skew) {
    const uint64_t pairs = (uint64_t)nq * nprobe;
    if (pairs && pairs * 2 <= ix->n_cus) sk_slices = (uint32_t)std::min<uint64_t>(dev_knob("MI355_LAT_SLICES_MAX", 8), 2 * ix->n_cus / pairs);
    sk_slices = std::max(1u, std::min(sk_slices, ix->max_len / 2048u));
    if (pl.kk > 256u) sk_slices = 1;  // (multi-pass selection re-scans per pass: keep whole partitions)
  }
  const uint32_t n_slices = skew ? sk_slices : std::max(1u, (ix->max_len + slice - 1) / slice);

  // chunk the batch so the workspace stays bounded
  const size_t spill_per_item = (size_t)(ix->m - m_lds) * 1024;  // table tail of one work item (k_scan_pair SPILL)
  const size_t per_q = (size_t)ix->nlist * 4 + (size_t)nprobe * n_slices * (pl.kk * sizeof(Cand) + spill_per_item);
  const size_t budget = (size_t)(pl.ws_mb ? pl.ws_mb : dev_knob("MI355_WORKSPACE_MB", 2048)) << 20;
  uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(nq, budget / std::max<size_t>(per_q, 1)));
  chunk = std::min(chunk, 65535u);  // grid.z limit
  if (skew) {
    ST_TRY(ix->items.ensure(sizeof(SkewItem) * (size_t)chunk * nprobe * n_slices));
    ST_TRY(ix->qthr.ensure(sizeof(uint32_t) * chunk));
    ST_TRY(ix->w_ccnt.ensure(sizeof(uint32_t) * (size_t)chunk * nprobe * n_slices));
  }
  ST_TRY(ix->w_qp.ensure(sizeof(float) * (size_t)chunk * ix->dim));
  ST_TRY(ix->w_qq.ensure(sizeof(float) * chunk));
  ST_TRY(ix->w_coarse.ensure(sizeof(float) * (size_t)chunk * ix->nlist));
  ST_TRY(ix->w_probes.ensure(sizeof(uint32_t) * (size_t)chunk * nprobe));
  ST_TRY(ix->w_cand.ensure(sizeof(Cand) * (size_t)chunk * nprobe * n_slices * pl.kk));
  if (spill_per_item) ST_TRY(ix->w_spill.ensure(spill_per_item * (size_t)chunk * nprobe * n_slices));
  // (refine: the ANN list and the exact list of a chunk; two sets when the refine of one call overlaps the next call's scan)
  const bool defer = pl.defer_refine && pl.refine && !pl.out_cand && chunk >= nq;
  if (pl.defer_refine && !defer) ST_TRY(join_exchange(ix));  // (a batch that needs several chunks keeps the serial path)
  // (each of the two sets has its own allocation: a set's layout depends on the shape of the call that uses it, and
  //  the other set may still be read by the re-rank of the call before — ADVICE round 3)
  const uint32_t rset = pl.rset & 1u;
  DevBuf& cand2 = rset ? ix->w_cand2b : ix->w_cand2;
  if (defer) {
    if (!ix->rstream) {
      HIP_TRY(hipStreamCreateWithFlags(&ix->rstream, hipStreamNonBlocking));
      for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipEventCreateWithFlags(&ix->r_scan[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ix->r_done[i], hipEventDisableTiming));
      }
    }
    if (ix->r_busy[rset]) HIP_TRY(hipStreamWaitEvent(st, ix->r_done[rset], 0));  // the refine two calls back: this set is free again
  }
  if (pl.refine && !pl.out_cand) {
    if (cand2.cap < sizeof(Cand) * (size_t)chunk * pl.kk * 2 && ix->r_busy[rset])
      HIP_TRY(hipEventSynchronize(ix->r_done[rset]));  // the set grows: its last user must be done before it is freed
    ST_TRY(cand2.ensure(sizeof(Cand) * (size_t)chunk * pl.kk * 2));
  }
  DevCtl* d_ctl = ix->w_ctl.as<DevCtl>();
  unsigned long long* d_stat = &d_ctl->rows_scanned;
  const bool prof = ix->profile != 0;

  for (uint32_t q0 = 0; q0 < nq; q0 += chunk) {
    const uint32_t n = std::min(chunk, nq - q0);
    const float* q = d_q + (size_t)q0 * ix->dim;
    ActiveMask act = pl.act;  // device-side batch size (second pass): this chunk's slots start at q0
    act.base += q0;
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
    // latency mode: a handful of queries run prep + coarse as one launch of single-wave workgroups
    const size_t small_lds = ((size_t)n * (((size_t)ix->dim + 3) & ~(size_t)3) + n) * sizeof(float);
    const bool small_front = !pl.ext_probes && !pl.act.n && n <= CS_MAXQ && small_lds <= 96u * 1024 && !dev_knob("MI355_COARSE_VALU", 0) &&
                             dev_knob("MI355_LAT_SMALL_FRONT", 1);
    if (!small_front)
      hipLaunchKernelGGL(k_prep_queries, dim3((n + 3) / 4), dim3(256), 4 * (((size_t)ix->dim + 3) & ~(size_t)3) * 4, st,
                         q, n, ix->dim, ix->metric, ix->w_qp.as<float>(), ix->w_qq.as<float>());
    if (pl.ext_probes) {
      // the probe list came from the two-phase coarse stage
      HIP_TRY(hipMemsetAsync(&d_ctl->bad_probes, 0, 4, st));
      const uint32_t np = n * nprobe;
      hipLaunchKernelGGL(k_take_probes, dim3((np + 255) / 256), dim3(256), 0, st, pl.ext_probes + (size_t)q0 * nprobe, np,
                         ix->nlist, view.plen, ix->w_probes.as<uint32_t>(), d_stat, &d_ctl->bad_probes, nprobe, act,
                         skew ? ix->qthr.as<uint32_t>() : (uint32_t*)nullptr);
      HIP_TRY(hipGetLastError());
      if (prof) HIP_TRY(hipEventRecord(es.ev[1], st));
    } else {
    if (small_front) {
      // lanes per centroid: the smallest split that gives the chip two waves per CU
      int lpc = 1;
      if ((ix->dim & 3u) == 0 && dev_knob("MI355_LAT_COARSE_SPLIT", 1))
        while (lpc < 16 && (uint64_t)ix->nlist * lpc < 2ull * 64 * ix->n_cus) lpc = lpc == 1 ? 4 : lpc * 2;
      if (lpc > 1) {
        const size_t lds = coarse_split_lds(n, ix->dim, lpc);
        auto go = [&](auto kern) -> int {
          if (lds > 48u * 1024) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          hipLaunchKernelGGL(kern, dim3((ix->nlist * lpc + 63) / 64), dim3(64), lds, st, q, n, ix->dim, ix->metric, view.centroids,
                             view.cnorm, ix->nlist, ix->w_qp.as<float>(), ix->w_qq.as<float>(), ix->w_coarse.as<float>());
          return MI355_OK;
        };
        const int rc = lpc == 4 ? go(k_coarse_split<4>) : lpc == 8 ? go(k_coarse_split<8>) : go(k_coarse_split<16>);
        if (rc != MI355_OK) return rc;
      } else {
      if (small_lds > 48u * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)k_coarse_small, hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds));
      hipLaunchKernelGGL(k_coarse_small, dim3((ix->nlist + 63) / 64), dim3(64), small_lds, st, q, n, ix->dim, ix->metric,
                         view.centroids, view.cnorm, ix->nlist, ix->w_qp.as<float>(), ix->w_qq.as<float>(),
                         ix->w_coarse.as<float>());
      }
    } else if (dev_knob("MI355_COARSE_VALU", 0))  // dev knob: the register-tiled VALU kernel (same bits)
      hipLaunchKernelGGL(k_coarse_tile, dim3((ix->nlist + CO_T - 1) / CO_T, (n + CO_T - 1) / CO_T),
                         dim3(256), 0, st, ix->w_qp.as<float>(), ix->w_qq.as<float>(), n,
                         view.centroids, view.cnorm, ix->nlist, ix->dim, ix->metric,
                         ix->w_coarse.as<float>());
    else
      hipLaunchKernelGGL(k_coarse_mfma, dim3((ix->nlist + CM_T - 1) / CM_T, (n + CM_T - 1) / CM_T),
                         dim3(256), 0, st, ix->w_qp.as<float>(), ix->w_qq.as<float>(), n,
                         view.centroids, view.cnorm, ix->nlist, ix->dim, ix->metric,
                         ix->w_coarse.as<float>(), act);
    HIP_TRY(hipGetLastError());
    if (prof) HIP_TRY(hipEventRecord(es.ev[1], st));
    hipLaunchKernelGGL(k_select_probes, dim3(n), dim3(256), 0, st, ix->w_coarse.as<float>(),
                       ix->nlist, nprobe, view.plen, ix->w_probes.as<uint32_t>(), d_stat, act,
                       skew ? ix->qthr.as<uint32_t>() : (uint32_t*)nullptr);
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
      pa.opos = ix->order.as<uint32_t>() + ix->nlist;
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
      pa.cand_cnt = ix->w_ccnt.as<uint32_t>();
      pa.kk = pl.kk;
      pa.nprobe = nprobe;
      // every query's nearest partition first: its kk-th best bounds the other partitions' admissions
      // (measured: scan -3 % at kk = 10, -5 % at kk = 64, -38 % at kk = 250 together with the block merge,
      // profiles/r03_g_*); an external probe list has its nearest partition at rank 0 when it comes from the
      // sharded coarse merge, otherwise rank 0 is just the caller's first probe
      pa.best_first = (pl.kk >= dev_knob("MI355_BEST_FIRST_MIN_KK", 1) && nprobe > 1u) ? 1u : 0u;
      pa.n_slices = n_slices;
      pa.act = act;
      const uint32_t pb = (pa.n_pairs + 255) / 256;
      // (qthr, the queries' running distance bounds, was reset by k_select_probes / k_take_probes)
      if (pa.n_pairs <= PLAN_SPARSE_MAX_PAIRS && dev_knob("MI355_PLAN_SPARSE", 1)) {
        hipLaunchKernelGGL(k_plan_sparse, dim3(1), dim3(PLAN_SPARSE_MAX_PAIRS), 0, st, pa);
      } else if (pa.n_pairs <= PLAN_FUSED_MAX_PAIRS && dev_knob("MI355_PLAN_FUSED", 1)) {
        hipLaunchKernelGGL(k_plan_fused, dim3(1), dim3(1024), 0, st, pa);
      } else {
        hipLaunchKernelGGL(k_plan_count, dim3(pb), dim3(256), 0, st, pa);
        hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(1024), 0, st, pa);
        hipLaunchKernelGGL(k_plan_fill, dim3(pb), dim3(256), 0, st, pa);
      }
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
      ka.cand_cnt = ix->w_ccnt.as<uint32_t>();
      ka.n_slices = n_slices;
      ka.dbg = dev_knob("MI355_DBG_SKIP", 0);
      ka.ctl = d_ctl;
      // a handle whose re-rank runs beside its scans (deferred refine over a host column) keeps a few CUs free for it:
      // a scan workgroup takes a whole CU (128 VGPRs x 16 waves), so nothing can share one with it
      const uint32_t scan_cus = (pl.defer_refine && ix->n_cus > 4 * MI355_REFINE_SIDE_CUS) ? ix->n_cus - MI355_REFINE_SIDE_CUS : ix->n_cus;
      uint32_t n_blocks = std::max(1u, (uint32_t)std::min<uint64_t>(scan_cus, (uint64_t)n * nprobe * n_slices));
      ka.n_slabs = ix->sk_slabs;
      ka.res_floats = ix->sk_res_floats;
      ka.partial = nullptr;
      ka.partial_stride = 0;
      if (ix->sk_slabs > 1) {
        // partial row sums between the slabs of a work item: 8 B per (tile position, unit, lane) of the longest partition,
        // per workgroup (persistent: one work item at a time)
        const uint64_t n_tiles = ((uint64_t)ix->max_len + SK_TILE - 1) / SK_TILE;
        const uint64_t stride = ((n_tiles + SK_STREAMS - 1) / SK_STREAMS + 1) * SK_UNITS * MI355_WAVE;
        if (stride >= (1ull << 31)) return fail(MI355_ERR_NOT_SUPPORTED, "a partition of %u rows is too long for the multi-slab scan", ix->max_len);
        while (n_blocks > 8 && stride * 8 * n_blocks > (2ull << 30)) n_blocks /= 2;  // (very long partitions: fewer workgroups)
        ST_TRY(ix->w_partial.ensure((size_t)stride * 8 * n_blocks));
        ka.partial = ix->w_partial.as<float2>();
        ka.partial_stride = (uint32_t)stride;
      }
      ST_TRY(launch_scan_skew(ka, ix->sk_M, ix->sk_slabbed, n_blocks, pl.kk, st));
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
      sa.dbg = dev_knob("MI355_DBG_SKIP", 0);
      sa.ctl = d_ctl;
      sa.m_lds = m_lds;
      sa.lut_spill = ix->w_spill.as<float>();
      sa.act = act;
      ST_TRY(launch_scan_pair(sa, dim3(n_slices, nprobe, n), st, vpt, nt));
    }
    if (prof) HIP_TRY(hipEventRecord(es.ev[3], st));

    if (!ix->merge_block_tried) {  // the block reduction's lists take more LDS than a kernel gets by default
      ix->merge_block_tried = true;
      ix->merge_block_ok =
          hipFuncSetAttribute((const void*)k_merge_cands<1, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MERGE_BLOCK_LDS) == hipSuccess &&
          hipFuncSetAttribute((const void*)k_merge_cands<2, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MERGE_BLOCK_LDS) == hipSuccess &&
          hipFuncSetAttribute((const void*)k_merge_cands<4, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MERGE_BLOCK_LDS) == hipSuccess;
      if (!ix->merge_block_ok) (void)hipGetLastError();
    }
    // the reduction of the work items' candidate slots: one wave per query; a handful of queries (whose items all ran at
    // once and all returned full lists) get a 16-wave block each, which cuts the slots to a short list first
    auto launch_merge = [&](int kpl, uint32_t nq_, hipStream_t s_, const MergeArgs& m_) {
      if (m_.src_cnt && nq_ <= (uint32_t)dev_knob("MI355_MERGE_BLOCK_MAX_NQ", 64) && ix->merge_block_ok)
        launch_by_kpl(kpl, k_merge_cands<1, 16>, k_merge_cands<2, 16>, k_merge_cands<4, 16>, dim3(nq_), dim3(1024), MERGE_BLOCK_LDS, s_, m_);
      else
        launch_by_kpl(kpl, k_merge_cands<1>, k_merge_cands<2>, k_merge_cands<4>, dim3(nq_), dim3(64), 0, s_, m_);
    };
    MergeArgs ma = merge_args_dense(ix->w_cand.as<Cand>(), nprobe * n_slices, pl.kk, n, pl.k);
    ma.ctl = d_ctl;
    ma.act = act;
    if (skew) {  // the scan's work items report how many of their kk slots they filled ([query][probe rank])
      ma.src_cnt = ix->w_ccnt.as<uint32_t>();
      ma.cnt_stride = 1;
      ma.cnt_q_stride = nprobe * n_slices;
    }
    if (pl.out_cand) {
      // sharded search: the kk best ANN records of this shard; refine runs after the cross-rank merge
      ma.k_out = pl.kk;
      ma.out_cand = pl.out_cand + (size_t)q0 * pl.kk;
      ma.out_cnt = d_cnt_ann + q0;
      launch_merge(kpl_kk, n, st, ma);
      HIP_TRY(hipGetLastError());
      if (prof) HIP_TRY(hipEventRecord(es.ev[4], st));
    } else if (!pl.refine) {
      ma.out_ids = d_ids + (size_t)q0 * pl.k;
      ma.out_dist = d_dist + (size_t)q0 * pl.k;
      ma.out_cnt = d_cnt + q0;
      launch_merge(kpl_k, n, st, ma);
      HIP_TRY(hipGetLastError());
      if (prof) HIP_TRY(hipEventRecord(es.ev[4], st));
    } else {
      // refine (query.rs:1313-1317): the kk ANN winners -> exact distances -> (distance, rowid) top k
      Cand* ann = cand2.as<Cand>();
      Cand* exact = ann + (size_t)chunk * pl.kk;
      ma.k_out = pl.kk;
      ma.out_cand = ann;
      ma.out_cnt = d_cnt_ann + q0;
      launch_merge(kpl_kk, n, st, ma);
      HIP_TRY(hipGetLastError());
      hipStream_t rs = st;
      if (defer) {  // the re-rank leaves the search stream: the next call's scan does not wait for it
        rs = ix->rstream;
        HIP_TRY(hipEventRecord(ix->r_scan[rset], st));
        HIP_TRY(hipStreamWaitEvent(rs, ix->r_scan[rset], 0));
      }
      if (prof) HIP_TRY(hipEventRecord(es.ev[4], rs));
      // (deferred: MI355_REFINE_SIDE_CUS workgroups in all — the CUs the scans of this handle leave free meanwhile)
      ST_TRY(launch_refine(ix, view, q, n, ann, d_cnt_ann + q0, nullptr, 0, pl.kk, pl.range, exact, rs, act,
                           defer ? std::max(1u, MI355_REFINE_SIDE_CUS / ((pl.kk + 255u) / 256u)) : 0u));
      MergeArgs mr = merge_args_dense(exact, 1, pl.kk, n, pl.k);
      mr.ctl = d_ctl;
      mr.act = act;
      mr.out_ids = d_ids + (size_t)q0 * pl.k;
      mr.out_dist = d_dist + (size_t)q0 * pl.k;
      mr.out_cnt = d_cnt + q0;
      launch_by_kpl(kpl_k, k_merge_cands<1>, k_merge_cands<2>, k_merge_cands<4>, dim3(n), dim3(64), 0, rs, mr);
      HIP_TRY(hipGetLastError());
      if (prof) {
        HIP_TRY(hipEventRecord(es.ev[5], rs));
        ix->ev_pending.push_back(es);
      }
      if (defer) {
        HIP_TRY(hipEventRecord(ix->r_done[rset], rs));
        ix->r_busy[rset] = true;
        if (!ix->xdone) HIP_TRY(hipEventCreateWithFlags(&ix->xdone, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ix->xdone, rs));
        ix->xpending = true;
      }
      continue;
    }
    if (prof) {
      HIP_TRY(hipEventRecord(es.ev[5], st));
      ix->ev_pending.push_back(es);
    }
  }
  ix->stats.work_items += (uint64_t)nq * nprobe * n_slices;
  return MI355_OK;
}

// ---- one call's launch sequence: control word, deadline, pipeline -------------------------------
// (everything here is stream work with stable arguments, so it can be captured in a hipGraph)
static int32_t launch_sequence(mi355_index* ix, const float* d_q, uint32_t nq, const SearchPlan& pl, uint64_t* d_ids,
                               float* d_dist, uint32_t* d_cnt, uint32_t* d_cnt_ann, uint32_t timeout_ms) {
  hipStream_t st = ix->stream;
  DevCtl* ctl = ix->w_ctl.as<DevCtl>();
  // (profile 2 = cumulative: the row counter runs until the next configure())
  hipLaunchKernelGGL(k_arm_deadline, dim3(1), dim3(1), 0, st, ctl, (unsigned long long)timeout_ms * ix->wall_khz,
                     (ix->profile & MI355_PROFILE_MASK) != 2 ? 1u : 0u);
  HIP_TRY(hipGetLastError());
  return run_ivfpq(ix, d_q, nq, pl, d_ids, d_dist, d_cnt, d_cnt_ann);
}

void account(mi355_index* ix, uint32_t nq, uint32_t nprobe) {
  ix->stats.n_queries += nq;
  ix->stats.partitions_probed += (uint64_t)nq * nprobe;
  ix->stats.scan_variant = ix->layout;
}

// Latency mode (MI355_CFG_GRAPH): small host-I/O batches replay a captured graph of the launch
// sequence (one submission instead of ~12 launches).  The first call with a given shape runs
// eagerly (it sizes the workspace); the second captures; a workspace re-allocation or different
// baked scalars re-capture.  Any capture failure leaves the shape on the eager path for good.
static int32_t run_graphed(mi355_index* ix, const float* d_q, uint32_t nq, const SearchPlan& pl, uint64_t* d_ids,
                           float* d_dist, uint32_t* d_cnt, uint32_t* d_cnt_ann, uint32_t timeout_ms, bool* used_graph) {
  *used_graph = false;
  GraphKey key{nq, pl.k, pl.kk, pl.nprobe,
               (pl.refine ? 1u : 0u) | (pl.range.has_lower ? 2u : 0u) | (pl.range.has_upper ? 4u : 0u)};
  GraphEntry& e = ix->graphs[key];
  const bool same = e.lower == pl.range.lower && e.upper == pl.range.upper && e.timeout_ms == timeout_ms &&
                    e.d_q == d_q && e.d_ids == d_ids;
  if (e.exec && e.gen == ix->ws_gen && same) {
    HIP_TRY(hipGraphLaunch(e.exec, ix->stream));
    ix->stats.work_items += e.work_items;
    ix->stats.graph_replays += 1;
    *used_graph = true;
    return MI355_OK;
  }
  if (e.exec) {
    (void)hipGraphExecDestroy(e.exec);
    e.exec = nullptr;
  }
  if (e.failed || !e.seen) {  // first sighting (or capture is known not to work): eager
    e.seen = true;
    return launch_sequence(ix, d_q, nq, pl, d_ids, d_dist, d_cnt, d_cnt_ann, timeout_ms);
  }
  hipGraph_t graph = nullptr;
  if (hipStreamBeginCapture(ix->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    e.failed = true;
    return launch_sequence(ix, d_q, nq, pl, d_ids, d_dist, d_cnt, d_cnt_ann, timeout_ms);
  }
  const uint64_t wi0 = ix->stats.work_items;
  const int32_t s = launch_sequence(ix, d_q, nq, pl, d_ids, d_dist, d_cnt, d_cnt_ann, timeout_ms);
  e.work_items = ix->stats.work_items - wi0;
  const hipError_t ce = hipStreamEndCapture(ix->stream, &graph);
  if (s != MI355_OK || ce != hipSuccess || !graph ||
      hipGraphInstantiate(&e.exec, graph, nullptr, nullptr, 0) != hipSuccess) {
    (void)hipGetLastError();
    if (graph) (void)hipGraphDestroy(graph);
    e.exec = nullptr;
    e.failed = true;
    ix->stats.work_items = wi0;
    return launch_sequence(ix, d_q, nq, pl, d_ids, d_dist, d_cnt, d_cnt_ann, timeout_ms);  // nothing ran yet
  }
  (void)hipGraphDestroy(graph);
  e.gen = ix->ws_gen;
  e.lower = pl.range.lower;
  e.upper = pl.range.upper;
  e.timeout_ms = timeout_ms;
  e.d_q = d_q;
  e.d_ids = d_ids;
  HIP_TRY(hipGraphLaunch(e.exec, ix->stream));
  ix->stats.graph_replays += 1;
  *used_graph = true;
  return MI355_OK;
}

struct SearchCall {  // one caller's buffers (host or device, per params->io_mem)
  const float* queries;
  uint32_t nq;
  uint64_t* out_rowids;
  float* out_dist;
  uint32_t* out_counts;
};

// host-side checks of a call (no device work); fills the derived numbers
int32_t check_search(mi355_index* ix, const float* queries, uint32_t n_queries, const mi355_search_params* p,
                            const uint64_t* ext_probes, uint32_t ext_nprobe, uint64_t* out_rowids, float* out_dist,
                            uint32_t* out_counts, SearchShape* sh, bool sharded_call) {
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
                "distance type %u does not match the metric the index was trained with (%u)", p->metric, ix->metric);
  if (n_queries && (!queries || !out_counts || (p->k && (!out_rowids || !out_dist))))
    return fail(MI355_ERR_INVALID_INPUT, "NULL query / output buffer");
  if (p->refine_factor && !ix->has_raw && !ix->raw_attached)
    return fail(MI355_ERR_INVALID_INPUT, "refine_factor needs raw vectors on the index handle");
  // neither `limit` (query.rs:818-907) nor `refine_factor` (query.rs:1302-1332) is bounded by the
  // reference; the only limit here is the 32-bit slot arithmetic of one query's candidate slots
  const uint64_t kk64 = (uint64_t)p->k * (p->refine_factor ? p->refine_factor : 1);
  if (kk64 > (1u << 20))
    return fail(MI355_ERR_NOT_SUPPORTED, "k * refine_factor = %llu exceeds 2^20 candidates per query",
                (unsigned long long)kk64);
  sh->k = p->k;
  sh->kk = (uint32_t)kk64;
  sh->np_min = std::min(p->nprobe_min, ix->nlist);
  sh->np_max = (p->nprobe_max == 0 || p->nprobe_max > ix->nlist) ? ix->nlist : p->nprobe_max;
  if (ext_probes) sh->np_min = sh->np_max = ext_nprobe;
  if (ix->shard_count > 1 && sh->np_max != sh->np_min && !sharded_call)
    return fail(MI355_ERR_NOT_SUPPORTED,
                "maximum_nprobes expansion on a sharded handle is decided after the cross-shard merge: use "
                "mi355_search_sharded");
  if ((uint64_t)sh->np_min * sh->kk > 0xFFFFFFFFull || (uint64_t)sh->np_max * sh->kk > 0xFFFFFFFFull)
    return fail(MI355_ERR_NOT_SUPPORTED, "nprobes * k * refine_factor exceeds 2^32 candidate slots per query");
  return MI355_OK;
}

// maximum_nprobes (query.rs:1246-1262): queries whose ANN stage found fewer than kk rows are searched
// again over np_max partitions; the decision is taken before the refine re-rank.  The short queries
// are picked ON THE DEVICE (k_compact_short) and the second pass runs over all n_queries slots behind
// an ActiveMask, so the host never reads the count: no synchronisation inside a device-I/O call, and
// in a sharded search no rank stalls the others.  `rows` receives [n_queries] slot -> query index and,
// behind them, the device-side count; `sq` the gathered query vectors.
int32_t expand_short_device(mi355_index* ix, const uint32_t* d_cnt_ann, uint32_t n_queries, uint32_t kk, const float* d_q,
                            DevBuf& rows, DevBuf& sq, hipStream_t st, ActiveMask* out_act) {
  ST_TRY(rows.ensure(sizeof(uint32_t) * ((size_t)n_queries + 1)));
  ST_TRY(sq.ensure(sizeof(float) * (size_t)n_queries * ix->dim));
  uint32_t* d_rows = rows.as<uint32_t>();
  uint32_t* d_n = d_rows + n_queries;
  hipLaunchKernelGGL(k_compact_short, dim3(1), dim3(1024), 0, st, d_cnt_ann, n_queries, kk, d_rows, d_n, ix->w_ctl.as<DevCtl>());
  ActiveMask act;
  act.n = d_n;
  act.base = 0;
  hipLaunchKernelGGL(k_gather_rows_f32, dim3(n_queries), dim3(256), 0, st, d_q, d_rows, ix->dim, sq.as<float>(), act);
  HIP_TRY(hipGetLastError());
  *out_act = act;
  return MI355_OK;
}

static int32_t expand_short_queries(mi355_index* ix, const float* d_q, uint32_t n_queries, const SearchPlan& pl,
                                    uint32_t np_max, uint64_t* d_ids, float* d_dist, uint32_t* d_cnt,
                                    const uint32_t* d_cnt_ann) {
  hipStream_t st = ix->stream;
  const uint32_t k = pl.k;
  ST_TRY(ix->w_sids.ensure(sizeof(uint64_t) * (size_t)n_queries * k));
  ST_TRY(ix->w_sdist.ensure(sizeof(float) * (size_t)n_queries * k));
  ST_TRY(ix->w_scnt.ensure(sizeof(uint32_t) * n_queries));
  ST_TRY(ix->w_scnt_ann.ensure(sizeof(uint32_t) * n_queries));
  SearchPlan p2 = pl;
  p2.nprobe = np_max;
  p2.ws_mb = 512;  // slots, not queries, size the workspace of this pass
  ST_TRY(expand_short_device(ix, d_cnt_ann, n_queries, pl.kk, d_q, ix->w_srows, ix->w_sq, st, &p2.act));
  ST_TRY(run_ivfpq(ix, ix->w_sq.as<float>(), n_queries, p2, ix->w_sids.as<uint64_t>(), ix->w_sdist.as<float>(),
                   ix->w_scnt.as<uint32_t>(), pl.refine ? ix->w_scnt_ann.as<uint32_t>() : ix->w_scnt.as<uint32_t>()));
  hipLaunchKernelGGL(k_scatter_results, dim3(n_queries), dim3(64), 0, st, ix->w_srows.as<uint32_t>(), k, ix->w_sids.as<uint64_t>(),
                     ix->w_sdist.as<float>(), ix->w_scnt.as<uint32_t>(), d_ids, d_dist, d_cnt, p2.act);
  HIP_TRY(hipGetLastError());
  ix->second_np = np_max;
  return MI355_OK;
}

int32_t join_exchange(mi355_index* ix) {
  if (ix->xpending) {
    HIP_TRY(hipStreamWaitEvent(ix->stream, ix->xdone, 0));
    ix->xpending = false;
    ix->r_busy[0] = ix->r_busy[1] = false;  // the search stream is now behind every deferred refine
  }
  return MI355_OK;
}

// The device work of one submission (ix->mu held): `calls` share `p`; host-I/O callers are packed
// into one device batch (the coalescing queue hands over several), device I/O is exactly one call.
static int32_t search_locked(mi355_index* ix, const std::vector<SearchCall>& calls, const mi355_search_params* p,
                             const SearchShape& sh, const uint64_t* ext_probes, uint32_t ext_nprobe) {
  HIP_TRY(hipSetDevice(ix->device));
  (void)hipGetLastError();  // the launch checks below must report THIS call's errors, not what another HIP user of the thread left
  const bool host_io = p->io_mem == MI355_MEM_HOST;
  // A device-I/O refine call without a deadline and without maximum_nprobes expansion leaves its exact re-rank on the
  // handle's refine stream (its outputs are complete at mi355_index_sync; the caller keeps queries and outputs
  // untouched until then, as for any device-I/O call): the NEXT call's scan starts at once.  With a host-mapped raw
  // column (C5) the re-rank is a PCIe gather, the scan an LDS / VALU loop: the two overlap almost entirely.
  const bool defer = !host_io && p->refine_factor != 0 && p->timeout_ms == 0 && sh.np_max == sh.np_min && !ext_probes &&
                     (ix->profile & MI355_PROFILE_MASK) != 1 && calls.size() == 1 && ix->raw_is_host && ix->defer_cfg;
  if (!defer) ST_TRY(join_exchange(ix));
  hipStream_t st = ix->stream;
  auto t_start = std::chrono::steady_clock::now();
  const uint32_t k = sh.k;
  uint32_t n_queries = 0;
  for (const SearchCall& c : calls) n_queries += c.nq;
  if ((ix->profile & MI355_PROFILE_MASK) != 2) {  // 2 = cumulative: counters run until the next configure()
    ST_TRY(drain_events(ix, true));
    const uint32_t replays = ix->stats.graph_replays;
    reset_stats(ix);
    ix->stats.graph_replays = replays;
  }
  ix->stats.coalesced_calls = (uint32_t)calls.size();

  const float* d_q = calls[0].queries;
  uint64_t* d_ids = calls[0].out_rowids;
  float* d_dist = calls[0].out_dist;
  uint32_t* d_cnt = calls[0].out_counts;
  // Small host batches (the latency path) travel through ONE page-locked staging block of the handle:
  // pageable hipMemcpyAsync stages (and, device-to-host, blocks) per call — four round trips for the
  // results of a single query.  Here: one H2D of the queries, the three result arrays carved out of one
  // device buffer and copied back by one D2H (+ the 64-byte control word), one synchronisation.
  const size_t q_bytes = sizeof(float) * (size_t)n_queries * ix->dim;
  const size_t r_bytes = (size_t)n_queries * k * (sizeof(uint64_t) + sizeof(float)) + sizeof(uint32_t) * (size_t)n_queries;
  const bool pinned = host_io && q_bytes + r_bytes <= ((size_t)4 << 20);
  unsigned char* h_pin = nullptr;
  if (host_io) {
    ST_TRY(ix->w_q.ensure(q_bytes));
    if (pinned) {
      const size_t need = q_bytes + r_bytes + sizeof(DevCtl) + 64;
      if (ix->h_pin_cap < need) {
        if (ix->h_pin) (void)hipHostFree(ix->h_pin);
        ix->h_pin = nullptr;
        ix->h_pin_cap = 0;
        HIP_TRY(hipHostMalloc(&ix->h_pin, need * 2, hipHostMallocDefault));
        ix->h_pin_cap = need * 2;
      }
      h_pin = (unsigned char*)ix->h_pin;
      ST_TRY(ix->w_ids.ensure(r_bytes));
      size_t off = 0;
      for (const SearchCall& c : calls) {
        memcpy(h_pin + off, c.queries, sizeof(float) * (size_t)c.nq * ix->dim);
        off += sizeof(float) * (size_t)c.nq * ix->dim;
      }
      HIP_TRY(hipMemcpyAsync(ix->w_q.p, h_pin, q_bytes, hipMemcpyHostToDevice, st));
      d_ids = ix->w_ids.as<uint64_t>();
      d_dist = (float*)(d_ids + (size_t)n_queries * k);
      d_cnt = (uint32_t*)(d_dist + (size_t)n_queries * k);
    } else {
      ST_TRY(ix->w_ids.ensure(sizeof(uint64_t) * (size_t)n_queries * k));
      ST_TRY(ix->w_dist.ensure(sizeof(float) * (size_t)n_queries * k));
      ST_TRY(ix->w_cnt.ensure(sizeof(uint32_t) * n_queries));
      uint32_t off = 0;
      for (const SearchCall& c : calls) {
        HIP_TRY(hipMemcpyAsync(ix->w_q.as<float>() + (size_t)off * ix->dim, c.queries, sizeof(float) * (size_t)c.nq * ix->dim,
                               hipMemcpyHostToDevice, st));
        off += c.nq;
      }
      d_ids = ix->w_ids.as<uint64_t>();
      d_dist = ix->w_dist.as<float>();
      d_cnt = ix->w_cnt.as<uint32_t>();
    }
    d_q = ix->w_q.as<float>();
  }
  SearchPlan pl;
  pl.k = k;
  pl.kk = sh.kk;
  pl.refine = p->refine_factor != 0;
  pl.nprobe = sh.np_min;
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
    pl.defer_refine = defer;
    pl.rset = defer ? (uint32_t)(ix->r_seq++ & 1u) : 0u;  // deferred calls alternate between two buffer sets
    DevBuf& cnt2 = pl.rset ? ix->w_cnt2b : ix->w_cnt2;
    if (cnt2.cap < sizeof(uint32_t) * n_queries && ix->r_busy[pl.rset]) HIP_TRY(hipEventSynchronize(ix->r_done[pl.rset]));
    ST_TRY(cnt2.ensure(sizeof(uint32_t) * n_queries));
    d_cnt_ann = cnt2.as<uint32_t>();
  }
  // latency mode: small host batches without profiling / prefilter / external probes replay a graph
  const bool graphable = ix->use_graph && host_io && n_queries <= 64 && (ix->profile & MI355_PROFILE_MASK) == 0 &&
                         !ext_probes && pl.filter.mode == MI355_FILTER_NONE;
  bool used_graph = false;
  if (graphable)
    ST_TRY(run_graphed(ix, d_q, n_queries, pl, d_ids, d_dist, d_cnt, d_cnt_ann, p->timeout_ms, &used_graph));
  else
    ST_TRY(launch_sequence(ix, d_q, n_queries, pl, d_ids, d_dist, d_cnt, d_cnt_ann, p->timeout_ms));
  account(ix, n_queries, pl.nprobe);

  if (sh.np_max > sh.np_min) ST_TRY(expand_short_queries(ix, d_q, n_queries, pl, sh.np_max, d_ids, d_dist, d_cnt, d_cnt_ann));

  if (host_io) {
    DevCtl h_ctl;
    if (pinned) {
      unsigned char* h_res = h_pin + q_bytes;
      DevCtl* h_c = (DevCtl*)(h_pin + ((q_bytes + r_bytes + 63) & ~(size_t)63));
      HIP_TRY(hipMemcpyAsync(h_res, d_ids, r_bytes, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(h_c, ix->w_ctl.p, sizeof(DevCtl), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      h_ctl = *h_c;
      const uint64_t* r_ids = (const uint64_t*)h_res;
      const float* r_dist = (const float*)(r_ids + (size_t)n_queries * k);
      const uint32_t* r_cnt = (const uint32_t*)(r_dist + (size_t)n_queries * k);
      uint32_t off = 0;
      for (const SearchCall& c : calls) {
        memcpy(c.out_rowids, r_ids + (size_t)off * k, sizeof(uint64_t) * (size_t)c.nq * k);
        memcpy(c.out_dist, r_dist + (size_t)off * k, sizeof(float) * (size_t)c.nq * k);
        memcpy(c.out_counts, r_cnt + off, sizeof(uint32_t) * c.nq);
        off += c.nq;
      }
    } else {
      uint32_t off = 0;
      for (const SearchCall& c : calls) {
        HIP_TRY(hipMemcpyAsync(c.out_rowids, d_ids + (size_t)off * k, sizeof(uint64_t) * (size_t)c.nq * k, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(c.out_dist, d_dist + (size_t)off * k, sizeof(float) * (size_t)c.nq * k, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(c.out_counts, d_cnt + off, sizeof(uint32_t) * c.nq, hipMemcpyDeviceToHost, st));
        off += c.nq;
      }
      HIP_TRY(hipMemcpyAsync(&h_ctl, ix->w_ctl.p, sizeof(DevCtl), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
    }
    ix->stats.timed_out = h_ctl.timed_out;
    ix->stats.bad_probes = h_ctl.bad_probes;
    // ids outside 0..nlist-1 are a caller error: report instead of returning a partial scan
    if (ext_probes && h_ctl.bad_probes)
      return fail(MI355_ERR_INVALID_INPUT, "%u probe ids are not partitions of this index", h_ctl.bad_probes);
    if (p->timeout_ms) {
      auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_start).count();
      if (h_ctl.timed_out || ms > (long long)p->timeout_ms)
        return fail(MI355_ERR_TIMEOUT, "Query timeout: %lld ms > %u ms%s", (long long)ms, p->timeout_ms,
                    h_ctl.timed_out ? " (stopped on the device)" : "");
    }
  }
  return MI355_OK;
}

static bool same_search(const mi355_search_params* a, const mi355_search_params* b) {
  return a->k == b->k && a->nprobe_min == b->nprobe_min && a->nprobe_max == b->nprobe_max &&
         a->refine_factor == b->refine_factor && a->metric == b->metric && a->has_lower_bound == b->has_lower_bound &&
         a->has_upper_bound == b->has_upper_bound && (!a->has_lower_bound || a->lower_bound == b->lower_bound) &&
         (!a->has_upper_bound || a->upper_bound == b->upper_bound) && a->timeout_ms == b->timeout_ms &&
         a->filter_mode == MI355_FILTER_NONE && b->filter_mode == MI355_FILTER_NONE && a->io_mem == b->io_mem;
}

// ext_probes != NULL: mi355_search_probes (the probe list replaces the coarse stage)
static int32_t search_impl(mi355_index* ix, const float* queries, uint32_t n_queries,
                           const mi355_search_params* p, const uint64_t* ext_probes, uint32_t ext_nprobe,
                           uint64_t* out_rowids, float* out_dist, uint32_t* out_counts) {
  SearchShape sh;
  ST_TRY(check_search(ix, queries, n_queries, p, ext_probes, ext_nprobe, out_rowids, out_dist, out_counts, &sh, false));
  if (n_queries == 0) return MI355_OK;
  if (sh.k == 0) {
    if (p->io_mem == MI355_MEM_HOST) memset(out_counts, 0, sizeof(uint32_t) * n_queries);
    else {
      std::lock_guard<std::mutex> lk(ix->mu);
      HIP_TRY(hipSetDevice(ix->device));
      HIP_TRY(hipMemsetAsync(out_counts, 0, sizeof(uint32_t) * n_queries, ix->stream));
    }
    return MI355_OK;
  }
  std::vector<SearchCall> calls{{queries, n_queries, out_rowids, out_dist, out_counts}};
  const bool queued = ix->coalesce && p->io_mem == MI355_MEM_HOST && !ext_probes && n_queries <= 256 &&
                      p->filter_mode == MI355_FILTER_NONE;
  if (!queued) {
    std::lock_guard<std::mutex> lk(ix->mu);
    return search_locked(ix, calls, p, sh, ext_probes, ext_nprobe);
  }
  // ---- coalescing queue (SURVEY.md §8b threading: callers are tokio workers, python/src/runtime.rs:31-37;
  // BaseTable: Send + Sync, table.rs:549).  A caller that finds the handle busy parks its request; the
  // thread that owns the device takes every parked request with the same parameters into ONE device
  // batch when it starts, so N concurrent single-query calls cost about one launch sequence.
  PendingSearch me;
  me.queries = queries;
  me.nq = n_queries;
  me.params = p;
  me.out_rowids = out_rowids;
  me.out_dist = out_dist;
  me.out_counts = out_counts;
  std::vector<PendingSearch*> served;
  {
    std::unique_lock<std::mutex> ql(ix->qmu);
    if (ix->busy) {
      ix->queue.push_back(&me);
      ix->qcv.wait(ql, [&] { return me.done || !ix->busy; });
      if (me.done) {
        if (me.status != MI355_OK) return fail(me.status, "%s", me.error.c_str());
        return MI355_OK;
      }
      ix->queue.erase(std::find(ix->queue.begin(), ix->queue.end(), &me));  // nobody served it: lead
    }
    ix->busy = true;
    uint32_t total = n_queries;
    for (auto it = ix->queue.begin(); it != ix->queue.end();) {
      if (same_search(p, (*it)->params) && total + (*it)->nq <= 4096) {
        total += (*it)->nq;
        served.push_back(*it);
        calls.push_back({(*it)->queries, (*it)->nq, (*it)->out_rowids, (*it)->out_dist, (*it)->out_counts});
        it = ix->queue.erase(it);
      } else {
        ++it;
      }
    }
  }
  int32_t status;
  {
    std::lock_guard<std::mutex> lk(ix->mu);
    status = search_locked(ix, calls, p, sh, nullptr, 0);
  }
  std::string err;
  if (status != MI355_OK) {
    char buf[600];
    mi355_last_error(buf, sizeof buf);
    err = buf;
  }
  {
    std::lock_guard<std::mutex> ql(ix->qmu);
    for (PendingSearch* f : served) {
      f->status = status;
      f->error = err;
      f->done = true;
    }
    ix->busy = false;
  }
  ix->qcv.notify_all();
  return status;
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

// the coarse stage over centroid slice [cent_lo, cent_hi) for device-resident queries (stream work only)
int32_t coarse_topn_device(mi355_index* ix, const float* d_q, uint32_t nq, uint32_t nprobe, uint32_t cent_lo,
                           uint32_t cent_hi, uint64_t* d_ids, float* d_dist, uint32_t* d_cnt) {
  hipStream_t st = ix->stream;
  const uint32_t n_slice = cent_hi - cent_lo, n_sel = std::min(nprobe, n_slice);
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
  return MI355_OK;
}

extern "C" int32_t mi355_coarse_topn(mi355_index* ix, const float* queries, uint32_t n_queries, uint32_t nprobe,
                                     uint32_t cent_lo, uint32_t cent_hi, uint32_t io_mem, uint64_t* out_part_ids,
                                     float* out_dist, uint32_t* out_counts) {
  if (!ix) return fail(MI355_ERR_INVALID_INPUT, "index is NULL");
  if (io_mem > MI355_MEM_DEVICE) return fail(MI355_ERR_INVALID_INPUT, "bad io_mem");
  if (cent_lo >= cent_hi || cent_hi > ix->nlist)
    return fail(MI355_ERR_INVALID_INPUT, "centroid slice [%u, %u) is not inside 0..%u", cent_lo, cent_hi, ix->nlist);
  if (nprobe == 0 || nprobe > ix->nlist) return fail(MI355_ERR_INVALID_INPUT, "nprobe must be in 1..nlist (%u)", ix->nlist);
  if (n_queries == 0) return MI355_OK;
  if (!queries || !out_part_ids || !out_dist || !out_counts) return fail(MI355_ERR_INVALID_INPUT, "NULL buffer");
  std::lock_guard<std::mutex> lk(ix->mu);
  HIP_TRY(hipSetDevice(ix->device));
  ST_TRY(join_exchange(ix));
  hipStream_t st = ix->stream;
  const bool host_io = io_mem == MI355_MEM_HOST;
  const uint32_t nq = n_queries;
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
  ST_TRY(coarse_topn_device(ix, d_q, nq, nprobe, cent_lo, cent_hi, d_ids, d_dist, d_cnt));
  if (host_io) {
    HIP_TRY(hipMemcpyAsync(out_part_ids, d_ids, sizeof(uint64_t) * (size_t)nq * nprobe, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_dist, d_dist, sizeof(float) * (size_t)nq * nprobe, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_counts, d_cnt, sizeof(uint32_t) * nq, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
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
  const int kpl = kpl_for(k);  // k > 256: passes of 256 rows
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
