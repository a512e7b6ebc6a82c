// ann_index_search.hip — the call driver of the IVF-PQ handle: request checks (nprobes validation,
// rust/lancedb/src/query.rs:1232-1275), the coalescing queue of concurrent host callers (python/src/runtime.rs:31-37),
// the hipGraph cache of small host batches, pinned staging, maximum_nprobes expansion, and the entry points
// mi355_search / mi355_search_probes / mi355_coarse_topn / mi355_merge_topk.  The pipeline it launches is run_ivfpq
// (ann_index.hip).
#include <thread>

#include "ann_internal.h"
#include "kernels_ivfpq.h"
#include "kernels_skew.h"

// ---- one call's launch sequence: control word, deadline, pipeline -------------------------------
// (everything here is stream work with stable arguments, so it can be captured in a hipGraph)
// `arm` = false: the control word was armed by the caller (graph replays arm it eagerly: the budget left differs from call to
// call — it shrinks by the time spent in the queue — and must not be baked into a captured graph)
static int32_t launch_sequence(mi355_index* ix, const float* d_q, uint32_t nq, const SearchPlan& pl, uint64_t* d_ids,
                               float* d_dist, uint32_t* d_cnt, uint32_t* d_cnt_ann, uint32_t timeout_ms, bool arm = true) {
  hipStream_t st = ix->stream;
  DevCtl* ctl = ix->w_ctl.as<DevCtl>();
  // (profile 2 = cumulative: the row counter runs until the next configure())
  const unsigned long long ticks = (unsigned long long)timeout_ms * ix->wall_khz;
  const uint32_t reset = (ix->profile & MI355_PROFILE_MASK) != 2 ? 1u : 0u;
  if (!arm) return run_ivfpq(ix, d_q, nq, pl, d_ids, d_dist, d_cnt, d_cnt_ann);
  if (lat_front_applies(ix, nq, pl)) {  // a handful of queries: the pipeline's first kernel arms the word (one launch less)
    SearchPlan pl2 = pl;
    pl2.arm_in_front = true;
    pl2.arm_ticks = ticks;
    pl2.arm_reset = reset;
    return run_ivfpq(ix, d_q, nq, pl2, d_ids, d_dist, d_cnt, d_cnt_ann);
  }
  hipLaunchKernelGGL(k_arm_deadline, dim3(1), dim3(1), 0, st, ctl, ticks, reset);
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
  const bool same = e.lower == pl.range.lower && e.upper == pl.range.upper && e.d_q == d_q && e.d_ids == d_ids;
  // the deadline is armed eagerly, in front of the replayed (or captured) sequence
  auto arm_now = [&]() -> int32_t {
    hipLaunchKernelGGL(k_arm_deadline, dim3(1), dim3(1), 0, ix->stream, ix->w_ctl.as<DevCtl>(),
                       (unsigned long long)timeout_ms * ix->wall_khz, (ix->profile & MI355_PROFILE_MASK) != 2 ? 1u : 0u);
    HIP_TRY(hipGetLastError());
    return MI355_OK;
  };
  if (e.exec && e.gen == ix->ws_gen && same) {
    ST_TRY(arm_now());
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
  ST_TRY(arm_now());
  if (hipStreamBeginCapture(ix->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    e.failed = true;
    return launch_sequence(ix, d_q, nq, pl, d_ids, d_dist, d_cnt, d_cnt_ann, timeout_ms);
  }
  const uint64_t wi0 = ix->stats.work_items;
  const int32_t s = launch_sequence(ix, d_q, nq, pl, d_ids, d_dist, d_cnt, d_cnt_ann, timeout_ms, /*arm=*/false);
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
  // when the call ENTERED the library: QueryExecutionOptions.timeout (query.rs:641) covers the whole call, the time it
  // waits in the coalescing queue and its batching window included
  std::chrono::steady_clock::time_point t0;
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
                                    const uint32_t* d_cnt_ann, bool host_io) {
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
  // A host-I/O call synchronises anyway: it reads the number of short queries and runs the second pass over exactly
  // those slots — none, in the common case (with maximum_nprobes = all partitions and a large batch the
  // device-side form walked thousands of mostly empty chunks).  Device-I/O calls keep the count on the device.
  uint32_t n2 = n_queries;
  if (host_io) {
    HIP_TRY(hipMemcpyAsync(&n2, p2.act.n, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    n2 = std::min(n2, n_queries);
    if (n2 == 0) {
      ix->second_np = np_max;
      return MI355_OK;
    }
  }
  ST_TRY(run_ivfpq(ix, ix->w_sq.as<float>(), n2, p2, ix->w_sids.as<uint64_t>(), ix->w_sdist.as<float>(),
                   ix->w_scnt.as<uint32_t>(), pl.refine ? ix->w_scnt_ann.as<uint32_t>() : ix->w_scnt.as<uint32_t>()));
  hipLaunchKernelGGL(k_scatter_results, dim3(n2), dim3(64), 0, st, ix->w_srows.as<uint32_t>(), k, ix->w_sids.as<uint64_t>(),
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
// `per_call` (coalesced batches): the outcome of every call of `all_calls` — MI355_OK, or MI355_ERR_TIMEOUT for a call whose
// OWN budget (QueryExecutionOptions.timeout is per query, rust/lancedb/src/query.rs:641) was spent before the device was reached
// or had passed when the batch completed; the return value is calls[0]'s, or a failure of the whole batch.
static int32_t search_locked(mi355_index* ix, const std::vector<SearchCall>& all_calls, const mi355_search_params* p,
                             const SearchShape& sh, const uint64_t* ext_probes, uint32_t ext_nprobe,
                             std::vector<int32_t>* per_call = nullptr) {
  HIP_TRY(hipSetDevice(ix->device));
  (void)hipGetLastError();  // the launch checks below must report THIS call's errors, not what another HIP user of the thread left
  const bool host_io = p->io_mem == MI355_MEM_HOST;
  // A device-I/O refine call without a deadline and without maximum_nprobes expansion leaves its exact re-rank on the
  // handle's refine stream (its outputs are complete at mi355_index_sync; the caller keeps queries and outputs
  // untouched until then, as for any device-I/O call): the NEXT call's scan starts at once.  With a host-mapped raw
  // column (C5) the re-rank is a PCIe gather, the scan an LDS / VALU loop: the two overlap almost entirely.
  const bool defer = !host_io && p->refine_factor != 0 && p->timeout_ms == 0 && sh.np_max == sh.np_min && !ext_probes &&
                     (ix->profile & MI355_PROFILE_MASK) != 1 && all_calls.size() == 1 && ix->raw_is_host && ix->defer_cfg;
  if (!defer) ST_TRY(join_exchange(ix));
  hipStream_t st = ix->stream;
  // Every call has its own deadline, counted from when IT entered the library (one expired call at the front of
  // the queue used to fail the whole batch, newcomers included — and the batch after it, and so on).  Calls whose budget is
  // already spent are failed here, one by one, and take no part in the batch; the device is armed with the smallest budget left
  // among the others.
  std::vector<SearchCall> live_calls;
  std::vector<size_t> live_at;  // index of each live call in all_calls
  uint32_t timeout_left = p->timeout_ms;
  if (per_call) per_call->assign(all_calls.size(), MI355_OK);
  if (p->timeout_ms) {
    std::vector<std::chrono::steady_clock::time_point> t0s;
    for (const SearchCall& c : all_calls) t0s.push_back(c.t0);
    const DeadlineSplit ds = split_by_deadline(t0s, std::chrono::steady_clock::now(), p->timeout_ms);  // (call_queue.h)
    timeout_left = ds.timeout_left;
    live_at = ds.live;
    if (per_call) {
      per_call->assign(all_calls.size(), MI355_ERR_TIMEOUT);
      for (size_t i : live_at) (*per_call)[i] = MI355_OK;
    }
    if (live_at.empty() || (!per_call && live_at.size() != all_calls.size()))
      return fail(MI355_ERR_TIMEOUT, "Query timeout: %lld ms > %u ms (before the device was reached)", ds.worst_wait_ms, p->timeout_ms);
    if (live_at.size() != all_calls.size())
      for (size_t i : live_at) live_calls.push_back(all_calls[i]);
  }
  const std::vector<SearchCall>& calls = live_calls.empty() ? all_calls : live_calls;
  if (live_at.empty())
    for (size_t i = 0; i < all_calls.size(); ++i) live_at.push_back(i);
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
  bool zero_copy_out = false, q_by_arg = false;
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
      // one query of a plain search: it rides in the first kernel's argument block (k_coarse_lat) — no staging copy
      q_by_arg = n_queries == 1 && calls.size() == 1 && ix->dim <= CL_ARG_FLOATS && p->refine_factor == 0 && sh.np_max == sh.np_min && !ext_probes &&
                 !ix->use_graph && (ix->profile & MI355_PROFILE_MASK) == 0 && p->filter_mode == MI355_FILTER_NONE && dev_knob("MI355_LAT_Q_IN_ARG", 1);
      if (!q_by_arg) HIP_TRY(hipMemcpyAsync(ix->w_q.p, h_pin, q_bytes, hipMemcpyHostToDevice, st));
      // The result arrays of a small batch are the page-locked block itself: the last kernel of the call stores its
      // (k ids + k distances + count) per query straight into host memory (posted PCIe writes, complete at the stream
      // synchronisation below) — the copy kernel that used to follow it was 4 us and a launch per call.
      zero_copy_out = n_queries <= 64 && dev_knob("MI355_LAT_ZERO_COPY_OUT", 1);
      d_ids = zero_copy_out ? (uint64_t*)(h_pin + q_bytes) : ix->w_ids.as<uint64_t>();
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
  if (q_by_arg) {
    pl.host_q = calls[0].queries;
    if (!lat_front_applies(ix, n_queries, pl)) {  // (the shape does not take the latency front after all: stage the query as usual)
      pl.host_q = nullptr;
      HIP_TRY(hipMemcpyAsync(ix->w_q.p, h_pin, q_bytes, hipMemcpyHostToDevice, st));
    }
  }
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
    ST_TRY(run_graphed(ix, d_q, n_queries, pl, d_ids, d_dist, d_cnt, d_cnt_ann, timeout_left, &used_graph));
  else
    ST_TRY(launch_sequence(ix, d_q, n_queries, pl, d_ids, d_dist, d_cnt, d_cnt_ann, timeout_left));
  account(ix, n_queries, pl.nprobe);

  if (sh.np_max > sh.np_min) ST_TRY(expand_short_queries(ix, d_q, n_queries, pl, sh.np_max, d_ids, d_dist, d_cnt, d_cnt_ann, host_io));

  if (host_io) {
    DevCtl h_ctl;
    if (pinned) {
      unsigned char* h_res = h_pin + q_bytes;
      DevCtl* h_c = (DevCtl*)(h_pin + ((q_bytes + r_bytes + 63) & ~(size_t)63));
      if (!zero_copy_out) HIP_TRY(hipMemcpyAsync(h_res, d_ids, r_bytes, hipMemcpyDeviceToHost, st));
      // the control word only says something for calls with a deadline or an external probe list
      const bool want_ctl = p->timeout_ms != 0 || ext_probes != nullptr;
      if (want_ctl) HIP_TRY(hipMemcpyAsync(h_c, ix->w_ctl.p, sizeof(DevCtl), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (want_ctl) h_ctl = *h_c; else memset(&h_ctl, 0, sizeof h_ctl);
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
      const auto now = std::chrono::steady_clock::now();
      long long first_late = -1;
      for (size_t j = 0; j < calls.size(); ++j) {  // every call against its own clock
        const long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(now - calls[j].t0).count();
        if (h_ctl.timed_out || ms > (long long)p->timeout_ms) {
          if (per_call) (*per_call)[live_at[j]] = MI355_ERR_TIMEOUT;
          if (first_late < 0 && (live_at[j] == 0 || !per_call)) first_late = ms;
        }
      }
      if (h_ctl.timed_out && per_call)  // (the batch is incomplete: nobody's results are)
        return fail(MI355_ERR_TIMEOUT, "Query timeout: the batch was stopped on the device after %u ms", timeout_left);
      if (first_late >= 0)
        return fail(MI355_ERR_TIMEOUT, "Query timeout: %lld ms > %u ms%s", first_late, p->timeout_ms,
                    h_ctl.timed_out ? " (stopped on the device)" : "");
    }
  }
  if (per_call && (*per_call)[0] == MI355_ERR_TIMEOUT)  // (calls[0] — the caller that runs the batch — was itself too late for it)
    return fail(MI355_ERR_TIMEOUT, "Query timeout: more than %u ms before the device was reached", p->timeout_ms);
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
  const auto t_entry = std::chrono::steady_clock::now();
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
  std::vector<SearchCall> calls{{queries, n_queries, out_rowids, out_dist, out_counts, t_entry}};
  const bool queued = ix->coalesce && p->io_mem == MI355_MEM_HOST && !ext_probes && n_queries <= 256 &&
                      p->filter_mode == MI355_FILTER_NONE;
  if (!queued) {
    std::lock_guard<std::mutex> lk(ix->mu);
    return search_locked(ix, calls, p, sh, ext_probes, ext_nprobe);
  }
  // ---- coalescing queue (call_queue.h): N concurrent single-query calls cost about one launch sequence
  PendingSearch me;
  me.queries = queries;
  me.nq = n_queries;
  me.params = p;
  me.out_rowids = out_rowids;
  me.out_dist = out_dist;
  me.out_counts = out_counts;
  me.t0 = t_entry;
  std::vector<PendingSearch*> served;
  if (!ix->cq.enter(me, [&](const PendingSearch& o) { return same_search(p, o.params); }, 4096u, served)) {
    if (me.status != MI355_OK) return fail(me.status, "%s", me.error);  // (the batch that carried it failed)
    return MI355_OK;
  }
  // This caller owns the device now and other callers sleep on its batch: whatever happens below — a failing status OR
  // a C++ exception on its way to the ABI guard — they must be released and the device handed on, so leave() runs
  // from a scope guard.
  struct LeaveGuard {
    mi355_index* ix;
    std::vector<PendingSearch*>& served;
    int32_t status = MI355_ERR_RUNTIME;
    char err[256] = "the batch that carried this call failed with a C++ exception in its leading call";
    ~LeaveGuard() { ix->cq.leave(served, status, err); }
  } guard{ix, served};
  for (PendingSearch* o : served) calls.push_back({o->queries, o->nq, o->out_rowids, o->out_dist, o->out_counts, o->t0});
  int32_t status;
  std::vector<int32_t> per_call;
  {
    std::lock_guard<std::mutex> lk(ix->mu);
    status = search_locked(ix, calls, p, sh, nullptr, 0, &per_call);
  }
  // A timeout is a call's own: a parked call whose outcome differs from this caller's gets it written here (the queue
  // delivers it instead of the batch's status).  Any other failure is the whole batch's.
  const bool batch_failed = status != MI355_OK && status != MI355_ERR_TIMEOUT;
  if (!batch_failed && per_call.size() == calls.size()) {
    const bool stopped = status == MI355_ERR_TIMEOUT && per_call[0] == MI355_OK;  // (the device stopped the batch: everybody timed out)
    for (size_t j = 0; j < served.size() && !stopped; ++j) {
      PendingSearch* o = served[j];
      o->status = per_call[j + 1];
      o->error[0] = 0;
      if (o->status == MI355_ERR_TIMEOUT) snprintf(o->error, sizeof o->error, "Query timeout: more than %u ms", p->timeout_ms);
      o->decided = true;
    }
  }
  guard.status = status;
  guard.err[0] = 0;
  if (status != MI355_OK) mi355_last_error(guard.err, sizeof guard.err);
  return status;
}

extern "C" int32_t mi355_search(mi355_index* ix, const float* queries, uint32_t n_queries,
                                const mi355_search_params* p, uint64_t* out_rowids,
                                float* out_dist, uint32_t* out_counts) try {
  return search_impl(ix, queries, n_queries, p, nullptr, 0, out_rowids, out_dist, out_counts);
} MI355_ABI_GUARD("mi355_search")

extern "C" int32_t mi355_search_probes(mi355_index* ix, const float* queries, uint32_t n_queries,
                                       const mi355_search_params* p, const uint64_t* probes, uint32_t nprobe,
                                       uint64_t* out_rowids, float* out_dist, uint32_t* out_counts) try {
  if (!probes) return fail(MI355_ERR_INVALID_INPUT, "probes is NULL");
  return search_impl(ix, queries, n_queries, p, probes, nprobe, out_rowids, out_dist, out_counts);
} MI355_ABI_GUARD("mi355_search_probes")

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
    // (the register-blocked kernel when the launch has two 128 x 128 tiles per CU: same bits, see kernels_ivfpq.h)
    if ((ix->dim & 3u) == 0 && (uint64_t)((n_slice + CM2_T - 1) / CM2_T) * ((ny + CM2_T - 1) / CM2_T) >= 2u * (uint64_t)ix->n_cus)
      hipLaunchKernelGGL(k_coarse_mfma2, dim3((n_slice + CM2_T - 1) / CM2_T, (ny + CM2_T - 1) / CM2_T), dim3(256), 0, st,
                         ix->w_qp.as<float>() + (size_t)y0 * ix->dim, ix->w_qq.as<float>() + y0, ny, cen, cn, n_slice,
                         ix->dim, ix->metric, ix->w_coarse.as<float>() + (size_t)y0 * n_slice);
    else
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
                                     float* out_dist, uint32_t* out_counts) try {
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
} MI355_ABI_GUARD("mi355_coarse_topn")
// ------------------------------------------------------------------- merge --
extern "C" int32_t mi355_merge_topk(int32_t device, void* hip_stream, const uint64_t* in_rowids,
                                    const float* in_dist, const uint32_t* in_counts,
                                    uint32_t n_lists, uint32_t n_queries, uint32_t k,
                                    uint64_t* out_rowids, float* out_dist, uint32_t* out_counts) try {
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
} MI355_ABI_GUARD("mi355_merge_topk")

