// ann_comm.hip — the multi-GPU exchange behind include/mi355_ann.h: RCCL over xGMI, no PyTorch.
//
// the IVF partition list shards across the GPUs of one node (one process per
// GPU); a query's result is the top-k of the union of its probed partitions, so every rank
// scans the probed partitions it owns and ONE all-gather of per-rank candidate lists + a k-way
// merge on every rank gives the unsharded result.  The reference has no collective at all
// (no communication backend) — this exchange is the engine's own.
//
// Exchange unit: a "slab" per rank = [B x kk 16-byte candidate records (distance, local position,
// rowid)] [B x u32 counts] [16-byte trailer: rows scanned], packed so that one ncclAllGather of
// bytes moves everything (the per-query payload is tiny — 160 B at k = 10 — so the exchange is
// latency-bound: one collective per exchange, not one per field).
//
//   mi355_search_sharded      ANN lists -> gather -> merge [-> owner-side refine -> gather -> merge]
//                             [-> maximum_nprobes second pass for queries that came back short]
//   MI355_SHARD_COARSE        + one gather of per-rank (partition, coarse distance) lists first
//   mi355_flat_search_sharded rows sharded across ranks, same gather + merge
#include <rccl/rccl.h>

#include "ann_internal.h"
#include "kernels_ivfpq.h"

struct mi355_comm {
  ncclComm_t comm = nullptr;
  uint32_t rank = 0, world = 1;
  int32_t device = 0;
  DevBuf send, recv, glist, gowner, gcnt, probes, send2, recv2, tmp_ids, tmp_dist, tmp_cnt, w_q, w_ids, w_dist, w_cnt,
      sq, sids, sdist, scnt, short_rows;
  std::mutex mu;
  mi355_comm_stats stats{};
  // where the last ANN exchange left every rank's trailer (scanned rows): read by mi355_comm_last_stats
  hipStream_t stat_stream = nullptr;
  size_t stat_slab_bytes = 0, stat_trailer_off = 0;
};

#define NCCL_TRY(expr)                                                                                         \
  do {                                                                                                         \
    ncclResult_t _r = (expr);                                                                                  \
    if (_r != ncclSuccess)                                                                                     \
      return fail(MI355_ERR_RUNTIME, "RCCL error %d (%s) at %s:%d: %s", (int)_r, ncclGetErrorString(_r), __FILE__, \
                  __LINE__, #expr);                                                                            \
  } while (0)

static_assert(MI355_COMM_ID_BYTES >= sizeof(ncclUniqueId), "unique id buffer too small");

extern "C" int32_t mi355_comm_unique_id(void* out_id) {
  if (!out_id) return fail(MI355_ERR_INVALID_INPUT, "out_id is NULL");
  ncclUniqueId id;
  NCCL_TRY(ncclGetUniqueId(&id));
  memset(out_id, 0, MI355_COMM_ID_BYTES);
  memcpy(out_id, &id, sizeof id);
  return MI355_OK;
}

extern "C" int32_t mi355_comm_create(const void* id, uint32_t rank, uint32_t world, int32_t device, mi355_comm** out) {
  if (!out) return fail(MI355_ERR_INVALID_INPUT, "out is NULL");
  *out = nullptr;
  if (!id) return fail(MI355_ERR_INVALID_INPUT, "id is NULL");
  if (world == 0 || rank >= world || world > MI355_MAX_RANKS)
    return fail(MI355_ERR_INVALID_INPUT, "rank %u / world %u out of range (world <= %d)", rank, world, MI355_MAX_RANKS);
  ST_TRY(need_device(device));
  mi355_comm* c = new (std::nothrow) mi355_comm();
  if (!c) return fail(MI355_ERR_RUNTIME, "out of host memory");
  c->rank = rank;
  c->world = world;
  c->device = device;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  ncclResult_t r = ncclCommInitRank(&c->comm, (int)world, uid, (int)rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(MI355_ERR_RUNTIME, "ncclCommInitRank(rank %u of %u) failed: %s", rank, world, ncclGetErrorString(r));
  }
  c->stats.struct_size = sizeof(mi355_comm_stats);
  c->stats.world = world;
  c->stats.rank = rank;
  *out = c;
  return MI355_OK;
}

extern "C" int32_t mi355_comm_destroy(mi355_comm* c) {
  if (!c) return MI355_OK;
  (void)hipSetDevice(c->device);
  for (DevBuf* b : {&c->send, &c->recv, &c->glist, &c->gowner, &c->gcnt, &c->probes, &c->send2, &c->recv2, &c->tmp_ids,
                    &c->tmp_dist, &c->tmp_cnt, &c->w_q, &c->w_ids, &c->w_dist, &c->w_cnt, &c->sq, &c->sids, &c->sdist,
                    &c->scnt, &c->short_rows})
    b->release();
  if (c->comm) {
    // nothing of this communicator may still be in flight when its resources go away (a caller
    // that exits right after the destroy would otherwise race RCCL's own teardown): drain the
    // device, finalize (flushes outstanding operations and stops the proxy), then destroy
    (void)hipDeviceSynchronize();
    (void)ncclCommFinalize(c->comm);
    (void)ncclCommDestroy(c->comm);
  }
  delete c;
  return MI355_OK;
}

extern "C" int32_t mi355_comm_last_stats(mi355_comm* c, mi355_comm_stats* out) {
  if (!c || !out) return fail(MI355_ERR_INVALID_INPUT, "NULL argument");
  if (out->struct_size != sizeof(mi355_comm_stats)) return fail(MI355_ERR_INVALID_INPUT, "mi355_comm_stats.struct_size mismatch");
  std::lock_guard<std::mutex> lk(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  if (c->stat_slab_bytes && c->recv.p) {
    // the trailers of the last ANN exchange: every rank's scanned rows (waits for that call)
    HIP_TRY(hipStreamSynchronize(c->stat_stream));
    unsigned long long sum = 0, mx = 0;
    for (uint32_t r = 0; r < c->world; ++r) {
      unsigned long long v = 0;
      HIP_TRY(hipMemcpy(&v, (unsigned char*)c->recv.p + c->stat_slab_bytes * r + c->stat_trailer_off, 8, hipMemcpyDeviceToHost));
      c->stats.rows_scanned[r] = v;
      sum += v;
      mx = std::max(mx, v);
    }
    c->stats.imbalance = sum ? (float)((double)mx * c->world / (double)sum) : 1.f;
  }
  *out = c->stats;
  out->struct_size = sizeof(mi355_comm_stats);
  return MI355_OK;
}

extern "C" int32_t mi355_coarse_slice(uint32_t nlist, uint32_t world, uint32_t rank, uint32_t* out_lo, uint32_t* out_hi) {
  if (!out_lo || !out_hi || world == 0 || rank >= world) return fail(MI355_ERR_INVALID_INPUT, "bad arguments");
  *out_lo = (uint32_t)(((uint64_t)nlist * rank) / world);
  *out_hi = (uint32_t)(((uint64_t)nlist * (rank + 1)) / world);
  return MI355_OK;
}

// ---- slab layout --------------------------------------------------------------------------------
struct Slab {
  size_t cand_bytes, cnt_bytes, bytes;  // bytes: whole slab, a multiple of 16
  Slab(uint32_t nq, uint32_t kk) {
    cand_bytes = sizeof(Cand) * (size_t)nq * kk;
    cnt_bytes = ((sizeof(uint32_t) * (size_t)nq + 15) / 16) * 16;
    bytes = cand_bytes + cnt_bytes + 16;
  }
  Cand* cand(void* base) const { return (Cand*)base; }
  uint32_t* cnt(void* base) const { return (uint32_t*)((unsigned char*)base + cand_bytes); }
  unsigned long long* trailer(void* base) const { return (unsigned long long*)((unsigned char*)base + cand_bytes + cnt_bytes); }
};

// one packed all-gather of every rank's slab, then the k-way merge of the gathered lists (the same
// on every rank).  `recv` holds world slabs afterwards.
static int32_t gather_slabs(mi355_comm* c, const Slab& sl, DevBuf& send, DevBuf& recv, hipStream_t st) {
  ST_TRY(recv.ensure(sl.bytes * c->world));
  NCCL_TRY(ncclAllGather(send.p, recv.p, sl.bytes, ncclChar, c->comm, st));
  c->stats.n_gathers += 1;
  c->stats.bytes_gathered += sl.bytes * c->world;
  return MI355_OK;
}

static MergeArgs merge_args_gathered(const Slab& sl, void* recv, uint32_t world, uint32_t nq, uint32_t kk_in, uint32_t k_out) {
  MergeArgs m = merge_args_dense(sl.cand(recv), world, kk_in, nq, k_out);
  m.src_stride = sl.bytes / sizeof(Cand);  // slabs are multiples of 16 bytes
  m.q_stride = kk_in;
  m.src_cnt = sl.cnt(recv);
  m.cnt_stride = sl.bytes / sizeof(uint32_t);
  return m;
}

// The ANN stage of `nq` device-resident queries over `nprobe` partitions on every rank, gathered and
// merged: g_list [nq, kk] (records keep the OWNER's local position), g_owner [nq, kk], g_cnt [nq].
static int32_t sharded_ann(mi355_index* ix, mi355_comm* c, const float* d_q, uint32_t nq, SearchPlan pl, uint32_t flags,
                           Cand* g_list, uint32_t* g_owner, uint32_t* g_cnt) {
  hipStream_t st = ix->stream;
  if (flags & MI355_SHARD_COARSE) {
    // phase 1: this rank's slice of the centroids -> (partition, distance) lists -> gather -> the global probe list
    uint32_t lo = 0, hi = 0;
    ST_TRY(mi355_coarse_slice(ix->nlist, c->world, c->rank, &lo, &hi));
    const uint32_t np = pl.nprobe;
    const Slab sl(nq, np);
    ST_TRY(c->send2.ensure(sl.bytes));
    ST_TRY(c->tmp_ids.ensure(sizeof(uint64_t) * (size_t)nq * np));
    ST_TRY(c->tmp_dist.ensure(sizeof(float) * (size_t)nq * np));
    HIP_TRY(hipMemsetAsync(sl.trailer(c->send2.p), 0, 16, st));
    if (hi > lo) {
      ST_TRY(coarse_topn_device(ix, d_q, nq, np, lo, hi, c->tmp_ids.as<uint64_t>(), c->tmp_dist.as<float>(), sl.cnt(c->send2.p)));
    } else {
      HIP_TRY(hipMemsetAsync(sl.cnt(c->send2.p), 0, sl.cnt_bytes, st));
    }
    hipLaunchKernelGGL(k_pack_cands, dim3((nq * np + 255) / 256), dim3(256), 0, st, c->tmp_ids.as<uint64_t>(),
                       c->tmp_dist.as<float>(), sl.cnt(c->send2.p), nq, np, sl.cand(c->send2.p));
    HIP_TRY(hipGetLastError());
    ST_TRY(gather_slabs(c, sl, c->send2, c->recv2, st));
    ST_TRY(c->probes.ensure(sizeof(uint64_t) * (size_t)nq * np));
    ST_TRY(c->tmp_cnt.ensure(sizeof(uint32_t) * nq));
    MergeArgs mp = merge_args_gathered(sl, c->recv2.p, c->world, nq, np, np);  // order: (distance, partition id)
    mp.out_ids = c->probes.as<uint64_t>();
    mp.out_cnt = c->tmp_cnt.as<uint32_t>();
    launch_by_kpl(kpl_for(np), k_merge_cands<1>, k_merge_cands<2>, k_merge_cands<4>, dim3(nq), dim3(64), 0, st, mp);
    HIP_TRY(hipGetLastError());
    pl.ext_probes = c->probes.as<uint64_t>();
  }
  const Slab sl(nq, pl.kk);
  ST_TRY(c->send.ensure(sl.bytes));
  pl.out_cand = sl.cand(c->send.p);
  ST_TRY(run_ivfpq(ix, d_q, nq, pl, nullptr, nullptr, nullptr, sl.cnt(c->send.p)));
  account(ix, nq, pl.nprobe);
  // trailer: this rank's scanned rows so far in this call (load imbalance report)
  HIP_TRY(hipMemcpyAsync(sl.trailer(c->send.p), &ix->w_ctl.as<DevCtl>()->rows_scanned, 8, hipMemcpyDeviceToDevice, st));
  ST_TRY(gather_slabs(c, sl, c->send, c->recv, st));
  MergeArgs ma = merge_args_gathered(sl, c->recv.p, c->world, nq, pl.kk, pl.kk);
  ma.out_cand = g_list;
  ma.out_owner = g_owner;
  ma.out_cnt = g_cnt;
  launch_by_kpl(kpl_for(pl.kk), k_merge_cands<1>, k_merge_cands<2>, k_merge_cands<4>, dim3(nq), dim3(64), 0, st, ma);
  HIP_TRY(hipGetLastError());
  c->stat_stream = st;
  c->stat_slab_bytes = sl.bytes;
  c->stat_trailer_off = sl.cand_bytes + sl.cnt_bytes;
  return MI355_OK;
}

// refine of the merged list: every rank scores the candidates it owns, a second gather + merge keeps k
static int32_t sharded_finish(mi355_index* ix, mi355_comm* c, const float* d_q, uint32_t nq, const SearchPlan& pl,
                              const Cand* g_list, const uint32_t* g_owner, const uint32_t* g_cnt, uint64_t* d_ids,
                              float* d_dist, uint32_t* d_cnt) {
  hipStream_t st = ix->stream;
  if (!pl.refine) {  // kk == k: the merged list is the result
    MergeArgs m = merge_args_dense(g_list, 1, pl.kk, nq, pl.k);
    m.out_ids = d_ids;
    m.out_dist = d_dist;
    m.out_cnt = d_cnt;
    launch_by_kpl(kpl_for(pl.k), k_merge_cands<1>, k_merge_cands<2>, k_merge_cands<4>, dim3(nq), dim3(64), 0, st, m);
    HIP_TRY(hipGetLastError());
    return MI355_OK;
  }
  const Slab sl(nq, pl.kk);
  ST_TRY(c->send2.ensure(sl.bytes));
  HIP_TRY(hipMemsetAsync(sl.trailer(c->send2.p), 0, 16, st));
  HIP_TRY(hipMemcpyAsync(sl.cnt(c->send2.p), g_cnt, sizeof(uint32_t) * nq, hipMemcpyDeviceToDevice, st));
  const IndexView view = make_view(ix);
  ST_TRY(launch_refine(ix, view, d_q, nq, g_list, g_cnt, g_owner, c->rank, pl.kk, pl.range, sl.cand(c->send2.p), st));
  ST_TRY(gather_slabs(c, sl, c->send2, c->recv2, st));
  MergeArgs m = merge_args_gathered(sl, c->recv2.p, c->world, nq, pl.kk, pl.k);
  m.out_ids = d_ids;
  m.out_dist = d_dist;
  m.out_cnt = d_cnt;
  launch_by_kpl(kpl_for(pl.k), k_merge_cands<1>, k_merge_cands<2>, k_merge_cands<4>, dim3(nq), dim3(64), 0, st, m);
  HIP_TRY(hipGetLastError());
  return MI355_OK;
}

extern "C" int32_t mi355_search_sharded(mi355_index* ix, mi355_comm* c, const float* queries, uint32_t n_queries,
                                        const mi355_search_params* p, uint32_t flags, uint64_t* out_rowids,
                                        float* out_dist, uint32_t* out_counts) {
  if (!c) return fail(MI355_ERR_INVALID_INPUT, "comm is NULL");
  if (flags & ~(uint32_t)MI355_SHARD_COARSE) return fail(MI355_ERR_INVALID_INPUT, "unknown flags 0x%x", flags);
  SearchShape sh;
  ST_TRY(check_search(ix, queries, n_queries, p, nullptr, 0, out_rowids, out_dist, out_counts, &sh, true));
  if (ix->shard_count != c->world || ix->shard_rank != c->rank)
    return fail(MI355_ERR_INVALID_INPUT, "index handle is shard %u of %u but the communicator is rank %u of %u",
                ix->shard_rank, ix->shard_count, c->rank, c->world);
  if (ix->device != c->device) return fail(MI355_ERR_INVALID_INPUT, "index and communicator live on different devices");
  if (n_queries == 0) return MI355_OK;
  std::lock_guard<std::mutex> lk(ix->mu);
  std::lock_guard<std::mutex> lc(c->mu);
  HIP_TRY(hipSetDevice(ix->device));
  hipStream_t st = ix->stream;
  const bool host_io = p->io_mem == MI355_MEM_HOST;
  const uint32_t k = sh.k;
  if (k == 0) {
    if (host_io) memset(out_counts, 0, sizeof(uint32_t) * n_queries);
    else HIP_TRY(hipMemsetAsync(out_counts, 0, sizeof(uint32_t) * n_queries, st));
    return MI355_OK;
  }
  auto t_start = std::chrono::steady_clock::now();
  if ((ix->profile & MI355_PROFILE_MASK) != 2) {
    ST_TRY(drain_events(ix, true));
    reset_stats(ix);
    HIP_TRY(hipMemsetAsync(&ix->w_ctl.as<DevCtl>()->rows_scanned, 0, 8, st));
  }
  hipLaunchKernelGGL(k_arm_deadline, dim3(1), dim3(1), 0, st, ix->w_ctl.as<DevCtl>(),
                     (unsigned long long)p->timeout_ms * ix->wall_khz);
  c->stats.n_gathers = 0;
  c->stats.bytes_gathered = 0;

  const float* d_q = queries;
  uint64_t* d_ids = out_rowids;
  float* d_dist = out_dist;
  uint32_t* d_cnt = out_counts;
  if (host_io) {
    ST_TRY(c->w_q.ensure(sizeof(float) * (size_t)n_queries * ix->dim));
    ST_TRY(c->w_ids.ensure(sizeof(uint64_t) * (size_t)n_queries * k));
    ST_TRY(c->w_dist.ensure(sizeof(float) * (size_t)n_queries * k));
    ST_TRY(c->w_cnt.ensure(sizeof(uint32_t) * n_queries));
    HIP_TRY(hipMemcpyAsync(c->w_q.p, queries, sizeof(float) * (size_t)n_queries * ix->dim, hipMemcpyHostToDevice, st));
    d_q = c->w_q.as<float>();
    d_ids = c->w_ids.as<uint64_t>();
    d_dist = c->w_dist.as<float>();
    d_cnt = c->w_cnt.as<uint32_t>();
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

  ST_TRY(c->glist.ensure(sizeof(Cand) * (size_t)n_queries * pl.kk));
  ST_TRY(c->gowner.ensure(sizeof(uint32_t) * (size_t)n_queries * pl.kk));
  ST_TRY(c->gcnt.ensure(sizeof(uint32_t) * n_queries));
  ST_TRY(sharded_ann(ix, c, d_q, n_queries, pl, flags, c->glist.as<Cand>(), c->gowner.as<uint32_t>(), c->gcnt.as<uint32_t>()));
  ST_TRY(sharded_finish(ix, c, d_q, n_queries, pl, c->glist.as<Cand>(), c->gowner.as<uint32_t>(), c->gcnt.as<uint32_t>(),
                        d_ids, d_dist, d_cnt));

  if (sh.np_max > sh.np_min) {
    // maximum_nprobes (query.rs:1246-1262): the merged ANN counts are identical on every rank, so all
    // ranks pick the same short queries and run the second pass (and its collectives) together
    std::vector<uint32_t> cnt(n_queries);
    HIP_TRY(hipMemcpyAsync(cnt.data(), c->gcnt.p, sizeof(uint32_t) * n_queries, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<uint32_t> shortq;
    for (uint32_t i = 0; i < n_queries; ++i)
      if (cnt[i] < pl.kk) shortq.push_back(i);
    if (!shortq.empty()) {
      const uint32_t ns = (uint32_t)shortq.size();
      ST_TRY(c->sq.ensure(sizeof(float) * (size_t)ns * ix->dim));
      ST_TRY(c->short_rows.ensure(sizeof(uint32_t) * ns));
      ST_TRY(c->sids.ensure(sizeof(uint64_t) * (size_t)ns * k));
      ST_TRY(c->sdist.ensure(sizeof(float) * (size_t)ns * k));
      ST_TRY(c->scnt.ensure(sizeof(uint32_t) * ns));
      HIP_TRY(hipMemcpyAsync(c->short_rows.p, shortq.data(), sizeof(uint32_t) * ns, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_gather_rows_f32, dim3(ns), dim3(256), 0, st, d_q, c->short_rows.as<uint32_t>(), ix->dim, c->sq.as<float>());
      HIP_TRY(hipGetLastError());
      SearchPlan p2 = pl;
      p2.nprobe = sh.np_max;
      ST_TRY(sharded_ann(ix, c, c->sq.as<float>(), ns, p2, flags, c->glist.as<Cand>(), c->gowner.as<uint32_t>(), c->gcnt.as<uint32_t>()));
      ST_TRY(sharded_finish(ix, c, c->sq.as<float>(), ns, p2, c->glist.as<Cand>(), c->gowner.as<uint32_t>(), c->gcnt.as<uint32_t>(),
                            c->sids.as<uint64_t>(), c->sdist.as<float>(), c->scnt.as<uint32_t>()));
      hipLaunchKernelGGL(k_scatter_results, dim3(ns), dim3(64), 0, st, c->short_rows.as<uint32_t>(), k, c->sids.as<uint64_t>(),
                         c->sdist.as<float>(), c->scnt.as<uint32_t>(), d_ids, d_dist, d_cnt);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(st));  // shortq (pageable host memory) was read by the upload
    }
  }

  if (host_io) {
    HIP_TRY(hipMemcpyAsync(out_rowids, d_ids, sizeof(uint64_t) * (size_t)n_queries * k, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_dist, d_dist, sizeof(float) * (size_t)n_queries * k, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_counts, d_cnt, sizeof(uint32_t) * n_queries, hipMemcpyDeviceToHost, st));
    DevCtl h_ctl;
    HIP_TRY(hipMemcpyAsync(&h_ctl, ix->w_ctl.p, sizeof(DevCtl), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    ix->stats.timed_out = h_ctl.timed_out;
    if (p->timeout_ms) {
      auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_start).count();
      if (h_ctl.timed_out || ms > (long long)p->timeout_ms)
        return fail(MI355_ERR_TIMEOUT, "Query timeout: %lld ms > %u ms", (long long)ms, p->timeout_ms);
    }
  }
  return MI355_OK;
}

// ---- flat, rows sharded across ranks -------------------------------------------------------------
extern "C" int32_t mi355_flat_search_sharded(mi355_flat* f, mi355_comm* c, const float* queries, uint32_t n_queries,
                                             const mi355_search_params* p, uint64_t* out_rowids, float* out_dist,
                                             uint32_t* out_counts) {
  if (!f || !c) return fail(MI355_ERR_INVALID_INPUT, "NULL handle");
  ST_TRY(validate_params(p));
  const uint32_t metric = p->metric == MI355_METRIC_DEFAULT ? (uint32_t)MI355_METRIC_L2 : p->metric;
  if (metric > MI355_METRIC_DOT) return fail(MI355_ERR_INVALID_INPUT, "unknown metric %u", metric);
  if (f->device != c->device) return fail(MI355_ERR_INVALID_INPUT, "flat handle and communicator live on different devices");
  if (n_queries == 0) return MI355_OK;
  if (!queries || !out_counts || (p->k && (!out_rowids || !out_dist)))
    return fail(MI355_ERR_INVALID_INPUT, "NULL query / output buffer");
  const uint32_t k = p->k;
  std::lock_guard<std::mutex> lk(f->mu);
  std::lock_guard<std::mutex> lc(c->mu);
  HIP_TRY(hipSetDevice(f->device));
  hipStream_t st = f->stream;
  const bool host_io = p->io_mem == MI355_MEM_HOST;
  if (k == 0) {
    if (host_io) memset(out_counts, 0, sizeof(uint32_t) * n_queries);
    else HIP_TRY(hipMemsetAsync(out_counts, 0, sizeof(uint32_t) * n_queries, st));
    return MI355_OK;
  }
  c->stats.n_gathers = 0;
  c->stats.bytes_gathered = 0;
  const float* d_q = queries;
  uint64_t* d_ids = out_rowids;
  float* d_dist = out_dist;
  uint32_t* d_cnt = out_counts;
  if (host_io) {
    ST_TRY(c->w_q.ensure(sizeof(float) * (size_t)n_queries * f->dim));
    ST_TRY(c->w_ids.ensure(sizeof(uint64_t) * (size_t)n_queries * k));
    ST_TRY(c->w_dist.ensure(sizeof(float) * (size_t)n_queries * k));
    ST_TRY(c->w_cnt.ensure(sizeof(uint32_t) * n_queries));
    HIP_TRY(hipMemcpyAsync(c->w_q.p, queries, sizeof(float) * (size_t)n_queries * f->dim, hipMemcpyHostToDevice, st));
    d_q = c->w_q.as<float>();
    d_ids = c->w_ids.as<uint64_t>();
    d_dist = c->w_dist.as<float>();
    d_cnt = c->w_cnt.as<uint32_t>();
  }
  // this rank's [B, k] result over its own rows, packed into the slab
  const Slab sl(n_queries, k);
  ST_TRY(c->send.ensure(sl.bytes));
  ST_TRY(c->tmp_ids.ensure(sizeof(uint64_t) * (size_t)n_queries * k));
  ST_TRY(c->tmp_dist.ensure(sizeof(float) * (size_t)n_queries * k));
  ST_TRY(run_flat_search_device(f, d_q, n_queries, p, c->tmp_ids.as<uint64_t>(), c->tmp_dist.as<float>(), sl.cnt(c->send.p)));
  hipLaunchKernelGGL(k_pack_cands, dim3((n_queries * k + 255) / 256), dim3(256), 0, st, c->tmp_ids.as<uint64_t>(),
                     c->tmp_dist.as<float>(), sl.cnt(c->send.p), n_queries, k, sl.cand(c->send.p));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemsetAsync(sl.trailer(c->send.p), 0, 16, st));
  ST_TRY(gather_slabs(c, sl, c->send, c->recv, st));
  MergeArgs m = merge_args_gathered(sl, c->recv.p, c->world, n_queries, k, k);
  m.out_ids = d_ids;
  m.out_dist = d_dist;
  m.out_cnt = d_cnt;
  launch_by_kpl(kpl_for(k), k_merge_cands<1>, k_merge_cands<2>, k_merge_cands<4>, dim3(n_queries), dim3(64), 0, st, m);
  HIP_TRY(hipGetLastError());
  if (host_io) {
    HIP_TRY(hipMemcpyAsync(out_rowids, d_ids, sizeof(uint64_t) * (size_t)n_queries * k, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_dist, d_dist, sizeof(float) * (size_t)n_queries * k, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_counts, d_cnt, sizeof(uint32_t) * n_queries, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  return MI355_OK;
}
