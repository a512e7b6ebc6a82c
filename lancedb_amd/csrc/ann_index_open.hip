// ann_index_open.hip — lifecycle of the IVF-PQ handle behind include/mi355_ann.h: open (ownership, code packing into the
// pre-skewed streams / generic blocks, planner tables), close, stream / configure / raw-column entry points.
// (split out of ann_index.hip in round 4; the search pipeline is ann_index.hip, the call driver ann_index_search.hip)
//
// Replaces the lance Session index cache (python/src/session.rs:49-50: upload once, reuse across queries); the
// arrays are IvfPqIndexBuilder's outputs (rust/lancedb/src/index/vector.rs:266-319).
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
                    &ix->w_partial,  &ix->w_cand2b,   &ix->w_cnt2b,    &ix->w_lutres, &ix->w_lutimg, &ix->w_tl};
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
                    &ix->w_filter, &ix->w_probes64, &ix->w_spill, &ix->w_partial, &ix->w_cand2b, &ix->w_cnt2b, &ix->w_lutres, &ix->w_lutimg})
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
    if (!force_pair && sk_shape(m, &shp)) {  // (4-bit codes are expanded to one byte per column at pack time)
      const uint32_t res_floats = shp.n_slabs > 1 ? shp.M * ix->dsub : d->dim;
      // what the packed streams cost against the source rows: padding to a 32-column tile, nibbles expanded to bytes
      // (m = 8 at 4 bits streams 8 x its code bytes, m = 1 thirty-two times) — past 8 x the generic layout is the
      // better index: it fits where this one may not and scans fewer bytes (8 x keeps every width from m = 4 up,
      // i.e. everything the reference's dim / 16 and dim / 8 rules produce, on the production scan)
      const uint64_t packed_row = (uint64_t)shp.M * shp.n_slabs, source_row = std::max<uint64_t>(1, ((uint64_t)m * d->nbits + 7) / 8);
      if (res_floats <= 2048 && sk_scan_lds(shp.M, res_floats, 8, 5) <= 160u * 1024 && packed_row <= 8 * source_row) {
        ix->layout = MI355_SCAN_SKEW;
        ix->sk_M = shp.M;
        ix->sk_slabs = shp.n_slabs;
        ix->sk_slabbed = shp.slabbed;
        ix->sk_res_floats = res_floats;
      }
    }
  }
  const bool skew = ix->layout == MI355_SCAN_SKEW;
  ix->lut_img_ok = lut_images_shape_ok(ix);  // batch-level distance tables (kernels_lut.h) for this shape
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
      sp.nbits = d->nbits;
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
    const size_t cb_elems = (size_t)m * cb_entries * ix->dsub;
    ST_TRY(ix->cbT.ensure(sizeof(float) * cb_elems));
    hipLaunchKernelGGL(k_transpose_codebook, dim3((uint32_t)((cb_elems + 255) / 256)), dim3(256), 0, st,
                       ix->codebook.as<float>(), m, ix->dsub, cb_entries, ix->cbT.as<float>());
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
    ST_TRY(ix->heads.ensure(sizeof(uint32_t) * 9 * SK_HEAD_STRIDE));  // 8 queue heads + the ticket word of k_select_plan (a line of its own)
    HIP_TRY(hipMemsetAsync(ix->heads.p, 0, sizeof(uint32_t) * 9 * SK_HEAD_STRIDE, st));
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

extern "C" int32_t mi355_index_open(const mi355_index_desc* desc, mi355_index** out) try {
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
} MI355_ABI_GUARD("mi355_index_open")

extern "C" int32_t mi355_index_close(mi355_index* index) try { return index_free(index); } MI355_ABI_GUARD("mi355_index_close")

extern "C" int32_t mi355_index_set_stream(mi355_index* ix, void* hip_stream) try {
  if (!ix) return fail(MI355_ERR_INVALID_INPUT, "index is NULL");
  std::lock_guard<std::mutex> lk(ix->mu);
  HIP_TRY(hipSetDevice(ix->device));
  // a deferred re-rank / overlapped exchange may still run on another stream: BOTH the old and the new search stream
  // wait for it (the next call on the new stream may reuse its buffers, and the caller's outputs must stay ordered
  // before later work on the stream the handle moves to — ADVICE round 4)
  const bool pending = ix->xpending;
  hipStream_t next = hip_stream ? (hipStream_t)hip_stream : ix->own_stream;
  if (pending && next != ix->stream) HIP_TRY(hipStreamWaitEvent(next, ix->xdone, 0));
  ST_TRY(join_exchange(ix));
  ix->stream = next;
  return MI355_OK;
} MI355_ABI_GUARD("mi355_index_set_stream")

extern "C" int32_t mi355_index_sync(mi355_index* ix) try {
  if (!ix) return fail(MI355_ERR_INVALID_INPUT, "index is NULL");
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t st;
  hipEvent_t xdone;
  {  // (the handle's stream and event are read under its lock: set_stream / a concurrent search may change them)
    std::lock_guard<std::mutex> lk(ix->mu);
    st = ix->stream;
    xdone = ix->xdone;
  }
  HIP_TRY(hipStreamSynchronize(st));
  if (xdone) HIP_TRY(hipEventSynchronize(xdone));  // an overlapped sharded search finishes on the communicator's stream
  // device-I/O calls cannot return their timeout: it is reported here (and in mi355_last_stats)
  uint32_t timed_out = 0;
  HIP_TRY(hipMemcpy(&timed_out, &ix->w_ctl.as<DevCtl>()->timed_out, 4, hipMemcpyDeviceToHost));
  if (timed_out) return fail(MI355_ERR_TIMEOUT, "Query timeout: the last search was stopped on the device");
  return MI355_OK;
} MI355_ABI_GUARD("mi355_index_sync")

extern "C" int32_t mi355_index_configure(mi355_index* ix, uint32_t scan_variant,
                                         uint32_t slice_rows, uint32_t profile) try {
  if (!ix) return fail(MI355_ERR_INVALID_INPUT, "index is NULL");
  if (scan_variant > MI355_SCAN_SKEW) return fail(MI355_ERR_INVALID_INPUT, "unknown scan variant");
  if ((profile & MI355_PROFILE_MASK) > 2 ||
      (profile & ~(uint32_t)(MI355_PROFILE_MASK | MI355_CFG_GRAPH | MI355_CFG_COALESCE | MI355_CFG_DEFER_REFINE | MI355_CFG_LUT_INLINE)))
    return fail(MI355_ERR_INVALID_INPUT, "unknown profile / mode bits 0x%x", profile);
  if (scan_variant != MI355_SCAN_AUTO && scan_variant != ix->layout)
    return fail(MI355_ERR_INVALID_INPUT,
                "scan variant %u does not match the code layout this index was packed for (%u)",
                scan_variant, ix->layout);
  std::lock_guard<std::mutex> lk(ix->mu);
  // quiesce first, commit the new mode only when that worked (a failing call leaves the handle as it was)
  HIP_TRY(hipSetDevice(ix->device));
  ST_TRY(join_exchange(ix));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  ST_TRY(drain_events(ix, true));
  HIP_TRY(hipMemset(ix->w_ctl.p, 0, sizeof(DevCtl)));
  // the ticket word of k_select_plan resets itself in the launch's last workgroup; a launch that never got there (aborted)
  // must not leave later calls reading a stale work list: every configure() starts from zero
  if (ix->heads.p) HIP_TRY(hipMemset(ix->heads.as<uint32_t>() + 8 * SK_HEAD_STRIDE, 0, sizeof(uint32_t)));
  reset_stats(ix);
  ix->scan_variant = scan_variant;
  ix->slice_rows = (slice_rows + 15u) & ~15u;
  ix->profile = profile & MI355_PROFILE_MASK;
  ix->use_graph = (profile & MI355_CFG_GRAPH) != 0;
  ix->coalesce.store((profile & MI355_CFG_COALESCE) != 0, std::memory_order_relaxed);
  ix->defer_cfg = (profile & MI355_CFG_DEFER_REFINE) != 0;
  ix->lut_inline_cfg = (profile & MI355_CFG_LUT_INLINE) != 0;
  ++ix->ws_gen;  // captured graphs bake in the slicing
  return MI355_OK;
} MI355_ABI_GUARD("mi355_index_configure")

extern "C" int32_t mi355_index_attach_raw(mi355_index* ix, const void* raw_vectors, uint32_t raw_dtype) try {
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
} MI355_ABI_GUARD("mi355_index_attach_raw")

extern "C" int32_t mi355_index_detach_raw(mi355_index* ix) try {
  if (!ix) return fail(MI355_ERR_INVALID_INPUT, "index is NULL");
  std::lock_guard<std::mutex> lk(ix->mu);
  HIP_TRY(hipSetDevice(ix->device));
  ST_TRY(join_exchange(ix));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  ix->raw_attached = nullptr;
  ix->raw_is_host = ix->raw_mapped_dev != nullptr;
  ++ix->ws_gen;
  return MI355_OK;
} MI355_ABI_GUARD("mi355_index_detach_raw")

extern "C" int32_t mi355_index_info(const mi355_index* ix, uint64_t* out_rows,
                                    uint32_t* out_partitions_owned) try {
  if (!ix) return fail(MI355_ERR_INVALID_INPUT, "index is NULL");
  if (out_rows) *out_rows = ix->n_local;
  if (out_partitions_owned) *out_partitions_owned = ix->parts_owned;
  return MI355_OK;
} MI355_ABI_GUARD("mi355_index_info")

