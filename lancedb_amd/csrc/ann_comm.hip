// ann_comm.hip — the multi-GPU exchange behind include/mi355_ann.h: RCCL over xGMI, no PyTorch.
//
// the IVF partition list shards across the GPUs of one node (one process per
// GPU); a query's result is the top-k of the union of its probed partitions, so every rank
// scans the probed partitions it owns and ONE all-gather of per-rank candidate lists + a k-way
// merge on every rank gives the unsharded result.  The reference has no collective at all
// (no communication backend) — this exchange is the engine's own.
//
// Exchange unit: a "slab" per rank = [B x kk 16-byte candidate records (distance, local position,
// rowid)] [B x u32 counts] [16-byte trailer: rows scanned, timed-out flag], packed so that one
// all-gather of bytes moves everything (the per-query payload is tiny — 160 B at k = 10 — so the
// exchange is latency-bound: one collective per exchange, not one per field).
//
//   mi355_search_sharded      ANN lists -> gather -> merge [-> owner-side refine -> gather -> merge]
//                             [-> maximum_nprobes second pass for queries that came back short,
//                                 picked on the device: no host synchronisation]
//   MI355_SHARD_COARSE        + one gather of per-rank (partition, coarse distance) lists first
//   mi355_flat_search_sharded rows sharded across ranks, same gather + merge
//
// Two transports carry the gather, everything above them is the same code:
//   * RCCL (mi355_comm_create): ncclAllGather, one process per GPU.  librccl is loaded lazily
//     (dlopen) by the first communicator call, so a single-GPU host needs no RCCL to search.
//   * loopback (mi355_comm_create_loopback): `world` ranks of ONE process on ONE device — each
//     rank is a caller thread with its own shard handle; the gather is a host rendezvous + device
//     copies between the ranks' slabs, stream-ordered by events.  It exists so that the world > 1
//     code (slab strides, owners, owner-side refine, the collective second pass, the sharded
//     coarse stage) runs against the unsharded oracle on a 1-GPU box, and to time the per-rank
//     stages of an N-rank step.
//
// Overlap (SURVEY.md §8e "overlap with the next batch's scan"): a device-I/O call without a
// timeout and without a second pass queues its exchange (gather, merge, owner-side refine, second
// gather, final merge) on the communicator's own stream behind an event; the next call's scan
// starts on the handle's stream right away.  Slab buffers are double-buffered per call parity.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <memory>

#include "ann_internal.h"
#include "kernels_ivfpq.h"

// ---- RCCL, loaded on first use ---------------------------------------------------------------------
namespace {
struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
  ncclResult_t (*CommFinalize)(ncclComm_t);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*CommAbort)(ncclComm_t);
  const char* (*GetErrorString)(ncclResult_t);
};
std::mutex g_rccl_mu;
RcclApi g_rccl{};
bool g_rccl_ok = false;
}  // namespace

static int32_t rccl_api(const RcclApi** out) {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (!g_rccl_ok) {
    // by SONAME first: a copy already mapped into the process (e.g. PyTorch's bundled one) is reused,
    // so there is never more than one collective library per process
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) return fail(MI355_ERR_RUNTIME, "RCCL is not available (dlopen librccl.so.1: %s)", dlerror());
    struct {
      const char* name;
      void** slot;
    } syms[] = {{"ncclGetUniqueId", (void**)&g_rccl.GetUniqueId},   {"ncclCommInitRank", (void**)&g_rccl.CommInitRank},
                {"ncclAllGather", (void**)&g_rccl.AllGather},       {"ncclCommFinalize", (void**)&g_rccl.CommFinalize},
                {"ncclCommDestroy", (void**)&g_rccl.CommDestroy},   {"ncclCommAbort", (void**)&g_rccl.CommAbort},
                {"ncclGetErrorString", (void**)&g_rccl.GetErrorString}};
    for (auto& s : syms) {
      *s.slot = dlsym(h, s.name);
      if (!*s.slot) return fail(MI355_ERR_RUNTIME, "librccl has no symbol %s", s.name);
    }
    g_rccl_ok = true;
  }
  *out = &g_rccl;
  return MI355_OK;
}

#define NCCL_TRY(api, expr)                                                                                     \
  do {                                                                                                          \
    ncclResult_t _r = (expr);                                                                                   \
    if (_r != ncclSuccess)                                                                                      \
      return fail(MI355_ERR_RUNTIME, "RCCL error %d (%s) at %s:%d: %s", (int)_r, (api)->GetErrorString(_r), __FILE__, \
                  __LINE__, #expr);                                                                             \
  } while (0)

static_assert(MI355_COMM_ID_BYTES >= sizeof(ncclUniqueId), "unique id buffer too small");

// ---- loopback transport ----------------------------------------------------------------------------
// The ranks of a loopback world meet at a host barrier inside every gather; a rank that fails in
// the collective part of a call aborts the group, so its peers return an error instead of waiting.
struct LoopGroup {
  std::mutex mu;
  std::condition_variable cv;
  uint32_t world = 0, arrived = 0;
  uint64_t gen = 0;
  bool aborted = false;
  std::vector<const void*> send;       // [world] this gather's source slab of every rank
  std::vector<hipEvent_t> ready;       // [world] recorded behind the producer of send[r]
  std::vector<hipEvent_t> copied;      // [world] recorded behind rank r's copies out of its peers' slabs
  ~LoopGroup() {
    for (auto& e : ready)
      if (e) (void)hipEventDestroy(e);
    for (auto& e : copied)
      if (e) (void)hipEventDestroy(e);
  }
  int32_t barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (aborted) return fail(MI355_ERR_RUNTIME, "loopback communicator was aborted by a failing rank");
    const uint64_t my = gen;
    if (++arrived == world) {
      arrived = 0;
      ++gen;
      cv.notify_all();
      return MI355_OK;
    }
    const bool ok = cv.wait_for(lk, std::chrono::seconds(120), [&] { return gen != my || aborted; });
    if (!ok || (aborted && gen == my)) {
      aborted = true;
      cv.notify_all();
      return fail(MI355_ERR_RUNTIME,
                  "loopback collective: %u of %u ranks arrived (every rank must call from its own thread)", arrived, world);
    }
    return MI355_OK;
  }
  void abort() {
    std::lock_guard<std::mutex> lk(mu);
    aborted = true;
    cv.notify_all();
  }
};

// buffers one call's exchange lives in; two sets alternate so that call i+1's scan may fill its
// slab while call i's exchange still reads the other one
struct XSet {
  DevBuf send, recv, glist, gowner, gcnt, send2, recv2;
  hipEvent_t scan_done = nullptr, done = nullptr;
  bool busy = false;  // `done` was recorded behind an exchange that may still run
};

struct mi355_comm {
  ncclComm_t comm = nullptr;
  const RcclApi* api = nullptr;
  std::shared_ptr<LoopGroup> loop;  // loopback transport (comm == nullptr)
  uint32_t rank = 0, world = 1;
  int32_t device = 0;
  bool dead = false;  // a collective failed half-way: the communicator cannot be used again
  XSet sets[2];
  uint64_t seq = 0;
  hipStream_t xstream = nullptr;  // the exchange of overlapped calls
  hipEvent_t x_begin = nullptr, x_end = nullptr;  // around the last call's exchange (stats)
  bool x_timed = false;
  DevBuf probes, tmp_ids, tmp_dist, tmp_cnt, w_q, w_ids, w_dist, w_cnt, sq, sids, sdist, scnt, short_rows;
  std::mutex mu;
  mi355_comm_stats stats{};
  // where the last ANN exchange left every rank's trailer: read by mi355_comm_last_stats
  hipStream_t stat_stream = nullptr;
  const void* stat_recv = nullptr;
  size_t stat_slab_bytes = 0, stat_trailer_off = 0;
};

struct SlabTrailer {
  unsigned long long rows_scanned;
  uint32_t timed_out;
  uint32_t pad;
};
static_assert(sizeof(SlabTrailer) == 16, "trailer is 16 bytes");

static __global__ void k_slab_trailer(const DevCtl* ctl, SlabTrailer* t) {
  t->rows_scanned = ctl ? ctl->rows_scanned : 0ull;
  t->timed_out = ctl ? ctl->timed_out : 0u;
  t->pad = 0;
}

static int32_t comm_common_init(mi355_comm* c) {
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
  for (XSet& s : c->sets) {
    HIP_TRY(hipEventCreateWithFlags(&s.scan_done, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  }
  HIP_TRY(hipEventCreate(&c->x_begin));
  HIP_TRY(hipEventCreate(&c->x_end));
  c->stats.struct_size = sizeof(mi355_comm_stats);
  c->stats.world = c->world;
  c->stats.rank = c->rank;
  return MI355_OK;
}

static void comm_free(mi355_comm* c) {
  (void)hipSetDevice(c->device);
  if (c->xstream) (void)hipStreamSynchronize(c->xstream);
  for (XSet& s : c->sets) {
    for (DevBuf* b : {&s.send, &s.recv, &s.glist, &s.gowner, &s.gcnt, &s.send2, &s.recv2}) b->release();
    if (s.scan_done) (void)hipEventDestroy(s.scan_done);
    if (s.done) (void)hipEventDestroy(s.done);
  }
  for (DevBuf* b : {&c->probes, &c->tmp_ids, &c->tmp_dist, &c->tmp_cnt, &c->w_q, &c->w_ids, &c->w_dist, &c->w_cnt, &c->sq,
                    &c->sids, &c->sdist, &c->scnt, &c->short_rows})
    b->release();
  if (c->x_begin) (void)hipEventDestroy(c->x_begin);
  if (c->x_end) (void)hipEventDestroy(c->x_end);
  if (c->comm && c->api) {
    // nothing of this communicator may still be in flight when its resources go away (a caller
    // that exits right after the destroy would otherwise race RCCL's own teardown): drain the
    // device, finalize (flushes outstanding operations and stops the proxy), then destroy
    (void)hipDeviceSynchronize();
    if (c->dead) {
      (void)c->api->CommAbort(c->comm);
    } else {
      (void)c->api->CommFinalize(c->comm);
      (void)c->api->CommDestroy(c->comm);
    }
  }
  if (c->xstream) (void)hipStreamDestroy(c->xstream);
  delete c;
}

extern "C" int32_t mi355_comm_unique_id(void* out_id) try {
  if (!out_id) return fail(MI355_ERR_INVALID_INPUT, "out_id is NULL");
  const RcclApi* api = nullptr;
  ST_TRY(rccl_api(&api));
  ncclUniqueId id;
  NCCL_TRY(api, api->GetUniqueId(&id));
  memset(out_id, 0, MI355_COMM_ID_BYTES);
  memcpy(out_id, &id, sizeof id);
  return MI355_OK;
} MI355_ABI_GUARD("mi355_comm_unique_id")

extern "C" int32_t mi355_comm_create(const void* id, uint32_t rank, uint32_t world, int32_t device, mi355_comm** out) try {
  if (!out) return fail(MI355_ERR_INVALID_INPUT, "out is NULL");
  *out = nullptr;
  if (!id) return fail(MI355_ERR_INVALID_INPUT, "id is NULL");
  if (world == 0 || rank >= world || world > MI355_MAX_RANKS)
    return fail(MI355_ERR_INVALID_INPUT, "rank %u / world %u out of range (world <= %d)", rank, world, MI355_MAX_RANKS);
  ST_TRY(need_device(device));
  const RcclApi* api = nullptr;
  ST_TRY(rccl_api(&api));
  mi355_comm* c = new (std::nothrow) mi355_comm();
  if (!c) return fail(MI355_ERR_RUNTIME, "out of host memory");
  c->rank = rank;
  c->world = world;
  c->device = device;
  c->api = api;
  int32_t s = comm_common_init(c);
  if (s != MI355_OK) {
    comm_free(c);
    return s;
  }
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  ncclResult_t r = api->CommInitRank(&c->comm, (int)world, uid, (int)rank);
  if (r != ncclSuccess) {
    c->comm = nullptr;
    comm_free(c);
    return fail(MI355_ERR_RUNTIME, "ncclCommInitRank(rank %u of %u) failed: %s", rank, world, api->GetErrorString(r));
  }
  *out = c;
  return MI355_OK;
} MI355_ABI_GUARD("mi355_comm_create")

extern "C" int32_t mi355_comm_create_loopback(uint32_t world, int32_t device, mi355_comm** out) try {
  if (!out) return fail(MI355_ERR_INVALID_INPUT, "out is NULL");
  for (uint32_t r = 0; r < world && r < MI355_MAX_RANKS; ++r) out[r] = nullptr;
  if (world == 0 || world > MI355_MAX_RANKS)
    return fail(MI355_ERR_INVALID_INPUT, "world %u out of range (1..%d)", world, MI355_MAX_RANKS);
  ST_TRY(need_device(device));
  auto g = std::make_shared<LoopGroup>();
  g->world = world;
  g->send.assign(world, nullptr);
  g->ready.assign(world, nullptr);
  g->copied.assign(world, nullptr);
  int32_t s = MI355_OK;
  for (uint32_t r = 0; r < world && s == MI355_OK; ++r) {
    if (hipEventCreateWithFlags(&g->ready[r], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g->copied[r], hipEventDisableTiming) != hipSuccess)
      s = fail(MI355_ERR_RUNTIME, "hipEventCreate failed");
  }
  for (uint32_t r = 0; r < world && s == MI355_OK; ++r) {
    mi355_comm* c = new (std::nothrow) mi355_comm();
    if (!c) {
      s = fail(MI355_ERR_RUNTIME, "out of host memory");
      break;
    }
    c->rank = r;
    c->world = world;
    c->device = device;
    c->loop = g;
    out[r] = c;
    s = comm_common_init(c);
  }
  if (s != MI355_OK) {
    for (uint32_t r = 0; r < world; ++r) {
      if (out[r]) comm_free(out[r]);
      out[r] = nullptr;
    }
  }
  return s;
} MI355_ABI_GUARD("mi355_comm_create_loopback")

extern "C" int32_t mi355_comm_destroy(mi355_comm* c) try {
  if (!c) return MI355_OK;
  if (c->loop) c->loop->abort();  // a peer still waiting for this rank must not wait for ever
  comm_free(c);
  return MI355_OK;
} MI355_ABI_GUARD("mi355_comm_destroy")

extern "C" int32_t mi355_comm_last_stats(mi355_comm* c, mi355_comm_stats* out) try {
  if (!c || !out) return fail(MI355_ERR_INVALID_INPUT, "NULL argument");
  if (out->struct_size != sizeof(mi355_comm_stats)) return fail(MI355_ERR_INVALID_INPUT, "mi355_comm_stats.struct_size mismatch");
  std::lock_guard<std::mutex> lk(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  if (c->stat_slab_bytes && c->stat_recv) {
    // the trailers of the last ANN exchange: every rank's scanned rows (waits for that call)
    HIP_TRY(hipStreamSynchronize(c->stat_stream));
    unsigned long long sum = 0, mx = 0;
    for (uint32_t r = 0; r < c->world; ++r) {
      SlabTrailer t{};
      HIP_TRY(hipMemcpy(&t, (const unsigned char*)c->stat_recv + c->stat_slab_bytes * r + c->stat_trailer_off, sizeof t,
                        hipMemcpyDeviceToHost));
      c->stats.rows_scanned[r] = t.rows_scanned;
      sum += t.rows_scanned;
      mx = std::max(mx, (unsigned long long)t.rows_scanned);
    }
    c->stats.imbalance = sum ? (float)((double)mx * c->world / (double)sum) : 1.f;
  } else {
    for (uint32_t r = 0; r < c->world; ++r) c->stats.rows_scanned[r] = 0;
    c->stats.imbalance = 1.f;
  }
  if (c->x_timed) {
    HIP_TRY(hipEventSynchronize(c->x_end));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->x_begin, c->x_end));
    c->stats.us_exchange = ms * 1000.f;
  }
  *out = c->stats;
  out->struct_size = sizeof(mi355_comm_stats);
  return MI355_OK;
} MI355_ABI_GUARD("mi355_comm_last_stats")

extern "C" int32_t mi355_coarse_slice(uint32_t nlist, uint32_t world, uint32_t rank, uint32_t* out_lo, uint32_t* out_hi) try {
  if (!out_lo || !out_hi || world == 0 || rank >= world) return fail(MI355_ERR_INVALID_INPUT, "bad arguments");
  *out_lo = (uint32_t)(((uint64_t)nlist * rank) / world);
  *out_hi = (uint32_t)(((uint64_t)nlist * (rank + 1)) / world);
  return MI355_OK;
} MI355_ABI_GUARD("mi355_coarse_slice")

// ---- slab layout --------------------------------------------------------------------------------
struct Slab {
  size_t cand_bytes, cnt_bytes, bytes;  // bytes: whole slab, a multiple of 16
  Slab(uint32_t nq, uint32_t kk) {
    cand_bytes = sizeof(Cand) * (size_t)nq * kk;
    cnt_bytes = ((sizeof(uint32_t) * (size_t)nq + 15) / 16) * 16;
    bytes = cand_bytes + cnt_bytes + sizeof(SlabTrailer);
  }
  Cand* cand(void* base) const { return (Cand*)base; }
  uint32_t* cnt(void* base) const { return (uint32_t*)((unsigned char*)base + cand_bytes); }
  SlabTrailer* trailer(void* base) const { return (SlabTrailer*)((unsigned char*)base + cand_bytes + cnt_bytes); }
};

// one packed all-gather of every rank's slab on stream `st`; `recv` (already sized) holds world
// slabs afterwards, rank r's at r * bytes
static int32_t gather_slabs(mi355_comm* c, const Slab& sl, const DevBuf& send, DevBuf& recv, hipStream_t st) {
  if (c->loop) {
    LoopGroup* g = c->loop.get();
    const uint32_t me = c->rank;
    // publish this rank's slab behind its producer, meet the peers, copy every slab on my stream
    HIP_TRY(hipEventRecord(g->ready[me], st));
    g->send[me] = send.p;
    ST_TRY(g->barrier());
    for (uint32_t r = 0; r < c->world; ++r) {
      if (r != me) HIP_TRY(hipStreamWaitEvent(st, g->ready[r], 0));
      HIP_TRY(hipMemcpyAsync((unsigned char*)recv.p + sl.bytes * r, g->send[r], sl.bytes, hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipEventRecord(g->copied[me], st));
    ST_TRY(g->barrier());
    // whatever this stream writes into `send` next must come after the peers' reads of it
    for (uint32_t r = 0; r < c->world; ++r)
      if (r != me) HIP_TRY(hipStreamWaitEvent(st, g->copied[r], 0));
  } else {
    NCCL_TRY(c->api, c->api->AllGather(send.p, recv.p, sl.bytes, ncclChar, c->comm, st));
  }
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

// every buffer of one pass over `nq` slots, sized BEFORE the first collective of the pass: an
// allocation failure then fails every rank alike instead of leaving the peers inside a gather
static int32_t ensure_pass_buffers(mi355_comm* c, XSet& xs, uint32_t nq, const SearchPlan& pl, uint32_t flags) {
  const Slab sl(nq, pl.kk);
  ST_TRY(xs.send.ensure(sl.bytes));
  ST_TRY(xs.recv.ensure(sl.bytes * c->world));
  ST_TRY(xs.glist.ensure(sizeof(Cand) * (size_t)nq * pl.kk));
  ST_TRY(xs.gowner.ensure(sizeof(uint32_t) * (size_t)nq * pl.kk));
  ST_TRY(xs.gcnt.ensure(sizeof(uint32_t) * nq));
  size_t s2 = pl.refine ? sl.bytes : 0;
  if (flags & MI355_SHARD_COARSE) {
    const Slab sp(nq, pl.nprobe);
    s2 = std::max(s2, sp.bytes);
    ST_TRY(c->tmp_ids.ensure(sizeof(uint64_t) * (size_t)nq * pl.nprobe));
    ST_TRY(c->tmp_dist.ensure(sizeof(float) * (size_t)nq * pl.nprobe));
    ST_TRY(c->probes.ensure(sizeof(uint64_t) * (size_t)nq * pl.nprobe));
    ST_TRY(c->tmp_cnt.ensure(sizeof(uint32_t) * nq));
  }
  if (s2) {
    ST_TRY(xs.send2.ensure(s2));
    ST_TRY(xs.recv2.ensure(s2 * c->world));
  }
  return MI355_OK;
}

// The ANN stage of `nq` device-resident queries over `nprobe` partitions on every rank, gathered and
// merged: xs.glist [nq, kk] (records keep the OWNER's local position), xs.gowner [nq, kk], xs.gcnt [nq].
// The scan runs on the handle's stream; the gather + merge on `sx` (the same stream, or the
// communicator's behind xs.scan_done).
static int32_t sharded_ann(mi355_index* ix, mi355_comm* c, XSet& xs, const float* d_q, uint32_t nq, SearchPlan pl,
                           uint32_t flags, hipStream_t sx) {
  hipStream_t st = ix->stream;
  if (flags & MI355_SHARD_COARSE) {
    // phase 1: this rank's slice of the centroids -> (partition, distance) lists -> gather -> the global probe list
    uint32_t lo = 0, hi = 0;
    ST_TRY(mi355_coarse_slice(ix->nlist, c->world, c->rank, &lo, &hi));
    const uint32_t np = pl.nprobe;
    const Slab sl(nq, np);
    hipLaunchKernelGGL(k_slab_trailer, dim3(1), dim3(1), 0, st, (const DevCtl*)nullptr, sl.trailer(xs.send2.p));
    if (hi > lo) {
      ST_TRY(coarse_topn_device(ix, d_q, nq, np, lo, hi, c->tmp_ids.as<uint64_t>(), c->tmp_dist.as<float>(), sl.cnt(xs.send2.p)));
    } else {
      HIP_TRY(hipMemsetAsync(sl.cnt(xs.send2.p), 0, sl.cnt_bytes, st));
    }
    hipLaunchKernelGGL(k_pack_cands, dim3((nq * np + 255) / 256), dim3(256), 0, st, c->tmp_ids.as<uint64_t>(),
                       c->tmp_dist.as<float>(), sl.cnt(xs.send2.p), nq, np, sl.cand(xs.send2.p));
    HIP_TRY(hipGetLastError());
    ST_TRY(gather_slabs(c, sl, xs.send2, xs.recv2, st));
    MergeArgs mp = merge_args_gathered(sl, xs.recv2.p, c->world, nq, np, np);  // order: (distance, partition id)
    mp.out_ids = c->probes.as<uint64_t>();
    mp.out_cnt = c->tmp_cnt.as<uint32_t>();
    mp.act = pl.act;
    launch_by_kpl(kpl_for(np), k_merge_cands<1>, k_merge_cands<2>, k_merge_cands<4>, dim3(nq), dim3(64), 0, st, mp);
    HIP_TRY(hipGetLastError());
    pl.ext_probes = c->probes.as<uint64_t>();
  }
  const Slab sl(nq, pl.kk);
  pl.out_cand = sl.cand(xs.send.p);
  ST_TRY(run_ivfpq(ix, d_q, nq, pl, nullptr, nullptr, nullptr, sl.cnt(xs.send.p)));
  account(ix, pl.act.n ? 0 : nq, pl.nprobe);
  // trailer: this rank's scanned rows so far in this call (load imbalance report) and its deadline flag
  hipLaunchKernelGGL(k_slab_trailer, dim3(1), dim3(1), 0, st, (const DevCtl*)ix->w_ctl.as<DevCtl>(), sl.trailer(xs.send.p));
  HIP_TRY(hipGetLastError());
  if (sx != st) {
    HIP_TRY(hipEventRecord(xs.scan_done, st));
    HIP_TRY(hipStreamWaitEvent(sx, xs.scan_done, 0));
  }
  HIP_TRY(hipEventRecord(c->x_begin, sx));
  ST_TRY(gather_slabs(c, sl, xs.send, xs.recv, sx));
  MergeArgs ma = merge_args_gathered(sl, xs.recv.p, c->world, nq, pl.kk, pl.kk);
  ma.out_cand = xs.glist.as<Cand>();
  ma.out_owner = xs.gowner.as<uint32_t>();
  ma.out_cnt = xs.gcnt.as<uint32_t>();
  ma.act = pl.act;
  launch_by_kpl(kpl_for(pl.kk), k_merge_cands<1>, k_merge_cands<2>, k_merge_cands<4>, dim3(nq), dim3(64), 0, sx, ma);
  HIP_TRY(hipGetLastError());
  c->stat_stream = sx;
  c->stat_recv = xs.recv.p;
  c->stat_slab_bytes = sl.bytes;
  c->stat_trailer_off = sl.cand_bytes + sl.cnt_bytes;
  return MI355_OK;
}

// refine of the merged list: every rank scores the candidates it owns, a second gather + merge keeps k
static int32_t sharded_finish(mi355_index* ix, mi355_comm* c, XSet& xs, const float* d_q, uint32_t nq, const SearchPlan& pl,
                              uint64_t* d_ids, float* d_dist, uint32_t* d_cnt, hipStream_t sx) {
  const Cand* g_list = xs.glist.as<Cand>();
  const uint32_t* g_owner = xs.gowner.as<uint32_t>();
  const uint32_t* g_cnt = xs.gcnt.as<uint32_t>();
  if (!pl.refine) {  // kk == k: the merged list is the result
    MergeArgs m = merge_args_dense(g_list, 1, pl.kk, nq, pl.k);
    m.out_ids = d_ids;
    m.out_dist = d_dist;
    m.out_cnt = d_cnt;
    m.act = pl.act;
    launch_by_kpl(kpl_for(pl.k), k_merge_cands<1>, k_merge_cands<2>, k_merge_cands<4>, dim3(nq), dim3(64), 0, sx, m);
    HIP_TRY(hipGetLastError());
  } else {
    const Slab sl(nq, pl.kk);
    hipLaunchKernelGGL(k_slab_trailer, dim3(1), dim3(1), 0, sx, (const DevCtl*)nullptr, sl.trailer(xs.send2.p));
    HIP_TRY(hipMemcpyAsync(sl.cnt(xs.send2.p), g_cnt, sizeof(uint32_t) * nq, hipMemcpyDeviceToDevice, sx));
    const IndexView view = make_view(ix);
    ST_TRY(launch_refine(ix, view, d_q, nq, g_list, g_cnt, g_owner, c->rank, pl.kk, pl.range, sl.cand(xs.send2.p), sx, pl.act));
    ST_TRY(gather_slabs(c, sl, xs.send2, xs.recv2, sx));
    MergeArgs m = merge_args_gathered(sl, xs.recv2.p, c->world, nq, pl.kk, pl.k);
    m.out_ids = d_ids;
    m.out_dist = d_dist;
    m.out_cnt = d_cnt;
    m.act = pl.act;
    launch_by_kpl(kpl_for(pl.k), k_merge_cands<1>, k_merge_cands<2>, k_merge_cands<4>, dim3(nq), dim3(64), 0, sx, m);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(c->x_end, sx));
  c->x_timed = true;
  return MI355_OK;
}

// the collective part of a sharded ANN call (a failure in here aborts the communicator)
static int32_t sharded_body(mi355_index* ix, mi355_comm* c, const float* queries, uint32_t n_queries,
                            const mi355_search_params* p, uint32_t flags, const SearchShape& sh, uint64_t* out_rowids,
                            float* out_dist, uint32_t* out_counts) {
  hipStream_t st = ix->stream;
  (void)hipGetLastError();  // report this call's errors only
  const bool host_io = p->io_mem == MI355_MEM_HOST;
  const uint32_t k = sh.k;
  const bool second_pass = sh.np_max > sh.np_min;
  // the exchange of a device-I/O call runs on the communicator's stream under the next call's scan
  // (not with a deadline armed: the next call would re-arm it; not with a second pass: it needs the merged counts)
  const bool overlap = !host_io && !second_pass && p->timeout_ms == 0 && !(flags & MI355_SHARD_NO_OVERLAP);
  hipStream_t sx = overlap ? c->xstream : st;
  XSet& xs = c->sets[c->seq & 1];
  ++c->seq;
  if (overlap) {
    if (xs.busy) HIP_TRY(hipStreamWaitEvent(st, xs.done, 0));  // the exchange two calls back: this set is free again
  } else {
    ST_TRY(join_exchange(ix));
    for (XSet& s : c->sets) s.busy = false;  // `st` is now behind every exchange
  }
  auto t_start = std::chrono::steady_clock::now();
  if ((ix->profile & MI355_PROFILE_MASK) != 2) {
    ST_TRY(drain_events(ix, true));
    reset_stats(ix);
    HIP_TRY(hipMemsetAsync(ix->w_ctl.p, 0, DEVCTL_COUNTER_BYTES, st));
  }
  hipLaunchKernelGGL(k_arm_deadline, dim3(1), dim3(1), 0, st, ix->w_ctl.as<DevCtl>(),
                     (unsigned long long)p->timeout_ms * ix->wall_khz);
  c->stats.n_gathers = 0;
  c->stats.bytes_gathered = 0;
  c->stats.overlapped = overlap ? 1u : 0u;

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

  SearchPlan p2 = pl;
  p2.nprobe = sh.np_max;
  p2.ws_mb = 512;  // slots, not queries, size the workspace of the second pass
  ST_TRY(ensure_pass_buffers(c, xs, n_queries, second_pass ? p2 : pl, flags));
  ST_TRY(ensure_pass_buffers(c, xs, n_queries, pl, flags));
  if (second_pass) {
    ST_TRY(c->sids.ensure(sizeof(uint64_t) * (size_t)n_queries * k));
    ST_TRY(c->sdist.ensure(sizeof(float) * (size_t)n_queries * k));
    ST_TRY(c->scnt.ensure(sizeof(uint32_t) * n_queries));
    ST_TRY(c->short_rows.ensure(sizeof(uint32_t) * ((size_t)n_queries + 1)));
    ST_TRY(c->sq.ensure(sizeof(float) * (size_t)n_queries * ix->dim));
  }

  ST_TRY(sharded_ann(ix, c, xs, d_q, n_queries, pl, flags, sx));
  ST_TRY(sharded_finish(ix, c, xs, d_q, n_queries, pl, d_ids, d_dist, d_cnt, sx));

  if (second_pass) {
    // maximum_nprobes (query.rs:1246-1262): the merged ANN counts are identical on every rank, so every
    // rank's device picks the same short queries and all ranks run the second pass (and its collectives)
    // together — over all n_queries slots behind the device-side count, no host round trip
    ST_TRY(expand_short_device(ix, xs.gcnt.as<uint32_t>(), n_queries, pl.kk, d_q, c->short_rows, c->sq, st, &p2.act));
    ST_TRY(sharded_ann(ix, c, xs, c->sq.as<float>(), n_queries, p2, flags, st));
    ST_TRY(sharded_finish(ix, c, xs, c->sq.as<float>(), n_queries, p2, c->sids.as<uint64_t>(), c->sdist.as<float>(),
                          c->scnt.as<uint32_t>(), st));
    hipLaunchKernelGGL(k_scatter_results, dim3(n_queries), dim3(64), 0, st, c->short_rows.as<uint32_t>(), k, c->sids.as<uint64_t>(),
                       c->sdist.as<float>(), c->scnt.as<uint32_t>(), d_ids, d_dist, d_cnt, p2.act);
    HIP_TRY(hipGetLastError());
    ix->second_np = sh.np_max;
  }

  if (overlap) {
    HIP_TRY(hipEventRecord(xs.done, sx));
    xs.busy = true;
    if (!ix->xdone) HIP_TRY(hipEventCreateWithFlags(&ix->xdone, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ix->xdone, sx));
    ix->xpending = true;
  }

  if (host_io) {
    HIP_TRY(hipMemcpyAsync(out_rowids, d_ids, sizeof(uint64_t) * (size_t)n_queries * k, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_dist, d_dist, sizeof(float) * (size_t)n_queries * k, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_counts, d_cnt, sizeof(uint32_t) * n_queries, hipMemcpyDeviceToHost, st));
    // the deadline verdict is the OR of every rank's flag (gathered in the trailers of the last ANN
    // exchange), so every rank returns the same status
    std::vector<SlabTrailer> tr(c->world);
    for (uint32_t r = 0; r < c->world; ++r)
      HIP_TRY(hipMemcpyAsync(&tr[r], (const unsigned char*)c->stat_recv + c->stat_slab_bytes * r + c->stat_trailer_off,
                             sizeof(SlabTrailer), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    uint32_t timed_out = 0;
    for (const SlabTrailer& t : tr) timed_out |= t.timed_out;
    ix->stats.timed_out = timed_out;
    if (p->timeout_ms && timed_out) {
      auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_start).count();
      return fail(MI355_ERR_TIMEOUT, "Query timeout: stopped on the device of a rank after %lld ms (limit %u ms)", (long long)ms,
                  p->timeout_ms);
    }
  }
  return MI355_OK;
}

extern "C" int32_t mi355_search_sharded(mi355_index* ix, mi355_comm* c, const float* queries, uint32_t n_queries,
                                        const mi355_search_params* p, uint32_t flags, uint64_t* out_rowids,
                                        float* out_dist, uint32_t* out_counts) try {
  if (!c) return fail(MI355_ERR_INVALID_INPUT, "comm is NULL");
  if (flags & ~(uint32_t)(MI355_SHARD_COARSE | MI355_SHARD_NO_OVERLAP)) return fail(MI355_ERR_INVALID_INPUT, "unknown flags 0x%x", flags);
  SearchShape sh;
  ST_TRY(check_search(ix, queries, n_queries, p, nullptr, 0, out_rowids, out_dist, out_counts, &sh, true));
  if (ix->shard_count != c->world || ix->shard_rank != c->rank)
    return fail(MI355_ERR_INVALID_INPUT, "index handle is shard %u of %u but the communicator is rank %u of %u",
                ix->shard_rank, ix->shard_count, c->rank, c->world);
  if (ix->device != c->device) return fail(MI355_ERR_INVALID_INPUT, "index and communicator live on different devices");
  if (n_queries == 0) return MI355_OK;
  std::lock_guard<std::mutex> lk(ix->mu);
  std::lock_guard<std::mutex> lc(c->mu);
  if (c->dead) return fail(MI355_ERR_RUNTIME, "the communicator failed inside an earlier collective and cannot be used again");
  HIP_TRY(hipSetDevice(ix->device));
  if (sh.k == 0) {
    if (p->io_mem == MI355_MEM_HOST) memset(out_counts, 0, sizeof(uint32_t) * n_queries);
    else HIP_TRY(hipMemsetAsync(out_counts, 0, sizeof(uint32_t) * n_queries, ix->stream));
    return MI355_OK;
  }
  const int32_t s = sharded_body(ix, c, queries, n_queries, p, flags, sh, out_rowids, out_dist, out_counts);
  if (s != MI355_OK && s != MI355_ERR_TIMEOUT) {
    // the peers may be inside a gather this rank will never join: fail them too
    c->dead = true;
    if (c->loop) c->loop->abort();
  }
  return s;
} MI355_ABI_GUARD("mi355_search_sharded")

// ---- flat, rows sharded across ranks -------------------------------------------------------------
static int32_t flat_sharded_body(mi355_flat* f, mi355_comm* c, const float* queries, uint32_t n_queries,
                                 const mi355_search_params* p, uint64_t* out_rowids, float* out_dist, uint32_t* out_counts) {
  hipStream_t st = f->stream;
  const uint32_t k = p->k;
  const bool host_io = p->io_mem == MI355_MEM_HOST;
  XSet& xs = c->sets[0];
  // serial on the flat handle's stream: behind any exchange a sharded ANN call left on the communicator's stream
  HIP_TRY(hipStreamSynchronize(c->xstream));
  for (XSet& s : c->sets) s.busy = false;
  c->stats.n_gathers = 0;
  c->stats.bytes_gathered = 0;
  c->stats.overlapped = 0;
  c->stat_slab_bytes = 0;  // the trailers of this exchange carry no scan counters
  c->stat_recv = nullptr;
  c->x_timed = false;
  const float* d_q = queries;
  uint64_t* d_ids = out_rowids;
  float* d_dist = out_dist;
  uint32_t* d_cnt = out_counts;
  const Slab sl(n_queries, k);
  ST_TRY(xs.send.ensure(sl.bytes));
  ST_TRY(xs.recv.ensure(sl.bytes * c->world));
  ST_TRY(c->tmp_ids.ensure(sizeof(uint64_t) * (size_t)n_queries * k));
  ST_TRY(c->tmp_dist.ensure(sizeof(float) * (size_t)n_queries * k));
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
  ST_TRY(run_flat_search_device(f, d_q, n_queries, p, c->tmp_ids.as<uint64_t>(), c->tmp_dist.as<float>(), sl.cnt(xs.send.p)));
  hipLaunchKernelGGL(k_pack_cands, dim3((n_queries * k + 255) / 256), dim3(256), 0, st, c->tmp_ids.as<uint64_t>(),
                     c->tmp_dist.as<float>(), sl.cnt(xs.send.p), n_queries, k, sl.cand(xs.send.p));
  hipLaunchKernelGGL(k_slab_trailer, dim3(1), dim3(1), 0, st, (const DevCtl*)nullptr, sl.trailer(xs.send.p));
  HIP_TRY(hipGetLastError());
  ST_TRY(gather_slabs(c, sl, xs.send, xs.recv, st));
  MergeArgs m = merge_args_gathered(sl, xs.recv.p, c->world, n_queries, k, k);
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

extern "C" int32_t mi355_flat_search_sharded(mi355_flat* f, mi355_comm* c, const float* queries, uint32_t n_queries,
                                             const mi355_search_params* p, uint64_t* out_rowids, float* out_dist,
                                             uint32_t* out_counts) try {
  if (!f || !c) return fail(MI355_ERR_INVALID_INPUT, "NULL handle");
  ST_TRY(validate_params(p));
  const uint32_t metric = p->metric == MI355_METRIC_DEFAULT ? (uint32_t)MI355_METRIC_L2 : p->metric;
  if (metric > MI355_METRIC_DOT) return fail(MI355_ERR_INVALID_INPUT, "unknown metric %u", metric);
  if (f->device != c->device) return fail(MI355_ERR_INVALID_INPUT, "flat handle and communicator live on different devices");
  if (n_queries == 0) return MI355_OK;
  if (!queries || !out_counts || (p->k && (!out_rowids || !out_dist)))
    return fail(MI355_ERR_INVALID_INPUT, "NULL query / output buffer");
  std::lock_guard<std::mutex> lk(f->mu);
  std::lock_guard<std::mutex> lc(c->mu);
  if (c->dead) return fail(MI355_ERR_RUNTIME, "the communicator failed inside an earlier collective and cannot be used again");
  HIP_TRY(hipSetDevice(f->device));
  if (p->k == 0) {
    if (p->io_mem == MI355_MEM_HOST) memset(out_counts, 0, sizeof(uint32_t) * n_queries);
    else HIP_TRY(hipMemsetAsync(out_counts, 0, sizeof(uint32_t) * n_queries, f->stream));
    return MI355_OK;
  }
  const int32_t s = flat_sharded_body(f, c, queries, n_queries, p, out_rowids, out_dist, out_counts);
  if (s != MI355_OK) {
    c->dead = true;
    if (c->loop) c->loop->abort();
  }
  return s;
} MI355_ABI_GUARD("mi355_flat_search_sharded")
