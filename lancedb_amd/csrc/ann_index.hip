// ann_index.hip — the IVF-PQ search PIPELINE behind include/mi355_ann.h: one pass of coarse quantiser -> probe select
// -> planner -> LUT build + ADC scan + top-k -> merge [-> refine -> merge] over a device-resident query batch
// (run_ivfpq), its stage timers and statistics.
//
// Replaces, behind lancedb::query::VectorQuery, what
// /root/reference/rust/lancedb/src/table/query.rs:219-327 hands to the lance
// Scanner (E1 in SURVEY.md §2b): nearest / nprobes / refine / distance_range /
// use_index -> a fixed launch sequence per query batch instead of a DataFusion
// plan.  Handle lifecycle: ann_index_open.hip; call driver (request checks, coalescing queue, graph cache, host I/O):
// ann_index_search.hip.
#include "ann_internal.h"
#include "kernels_ivfpq.h"
#include "kernels_skew.h"

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
      // ev[6] sits between the planner (+ the batch's distance-table images) and the scan kernel: us_scan is the scan
      // kernel's own time, us_plan what ran between the probe selection and it
      float ms_plan = 0;
      HIP_TRY(hipEventElapsedTime(&ms_plan, es.ev[2], es.ev[6]));
      const float us_plan = std::min(ms_plan * 1000.f, us[2]);
      ix->stats.us_coarse += us[0];
      ix->stats.us_select += us[1];
      ix->stats.us_plan += us_plan;
      ix->stats.us_scan += us[2] - us_plan;
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

extern "C" int32_t mi355_last_stats(mi355_index* ix, mi355_stats* out) try {
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
} MI355_ABI_GUARD("mi355_last_stats")

#if defined(MI355_DEV_COUNTERS) || defined(MI355_DEV_FRONT) || defined(MI355_DEV_PLAN)
// dev builds only (never in the product library): the scan's phase ticks and selection counters (or the front kernels' stage ticks)
extern "C" int32_t mi355_dev_counters(mi355_index* ix, uint32_t* out8, int32_t reset) try {
  HIP_TRY(hipSetDevice(ix->device));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  DevCtl h;
  HIP_TRY(hipMemcpy(&h, ix->w_ctl.p, sizeof h, hipMemcpyDeviceToHost));
  memcpy(out8, h.dev, sizeof h.dev);
  if (reset) HIP_TRY(hipMemset(ix->w_ctl.as<DevCtl>()->dev, 0, sizeof h.dev));
  return MI355_OK;
} MI355_ABI_GUARD("mi355_dev_counters")
#endif

#ifdef MI355_DEV_TIMELINE
// dev builds only: the per-workgroup stamps of the last scan launch (kernels_skew.h SK_TL) -> out[grid][SK_TL_WORDS]
extern "C" int32_t mi355_dev_timeline(mi355_index* ix, unsigned long long* out, uint32_t max_words, uint32_t* grid, uint32_t* words) try {
  HIP_TRY(hipSetDevice(ix->device));
  HIP_TRY(hipStreamSynchronize(ix->stream));
  *grid = ix->tl_grid;
  *words = SK_TL_WORDS;
  const size_t n = std::min<size_t>(max_words, (size_t)ix->tl_grid * SK_TL_WORDS);
  if (n) HIP_TRY(hipMemcpy(out, ix->w_tl.p, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return MI355_OK;
} MI355_ABI_GUARD("mi355_dev_timeline")
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

// CUs the scan's persistent workgroups take: a handle whose re-rank runs beside its scans (deferred refine over a host column)
// keeps a few CUs free for it — a scan workgroup takes a whole CU (128 VGPRs x 16 waves), so nothing can share one with it
static uint32_t scan_cus_of(const mi355_index* ix, const SearchPlan& pl) {
  return (pl.defer_refine && ix->n_cus > 4 * MI355_REFINE_SIDE_CUS) ? ix->n_cus - MI355_REFINE_SIDE_CUS : ix->n_cus;
}

bool lat_front_applies(const mi355_index* ix, uint32_t nq, const SearchPlan& pl) {
  const size_t small_lds = ((size_t)nq * (((size_t)ix->dim + 3) & ~(size_t)3) + nq) * sizeof(float);
  return ix->layout == MI355_SCAN_SKEW && !pl.ext_probes && !pl.act.n && nq >= 1 && nq <= CS_MAXQ && small_lds <= 96u * 1024 &&
         (ix->dim & 3u) == 0 && ix->nlist <= SELPLAN_MAX_NLIST && pl.nprobe <= ix->nlist &&
         (uint64_t)nq * pl.nprobe <= PLAN_SPARSE_MAX_PAIRS && !dev_knob("MI355_COARSE_VALU", 0) && dev_knob("MI355_LAT_SMALL_FRONT", 1) &&
         dev_knob("MI355_LAT_FRONT", 1);
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
  uint32_t sk_by_rows = 0;  // (PlanArgs::by_rows) the sparse planner cuts pairs by rows: sk_slices is then the most a pair is cut into
  if (skew) {
    const uint64_t pairs = (uint64_t)nq * nprobe;
    // (round 4: up to 3 work items per CU.  A batch of 8 queries is 512 whole-partition items on 256 CUs: its scan took
    //  370 us because the longest partition decides; cut in two it is bounded by half of it.  Every slice rebuilds the
    //  distance table, so batches that already give a CU 3 items keep whole partitions.)
    if (pairs && pairs < 3ull * ix->n_cus) {
      // (round 6) batches the sparse planner lays out are cut by rows, not by count: equal work items whatever the partition
      // lengths are, one per CU for a single query; a pair's candidate slots are strided by the most slices a pair may get
      // (such a batch builds its tables in the work items even where batch-level table images exist — below: two more launches
      //  cost a single query more than they save, and the image kernels walk a fixed number of items per pair)
      const bool sparse = pairs <= PLAN_SPARSE_MAX_PAIRS && dev_knob("MI355_PLAN_SPARSE", 1) && dev_knob("MI355_LAT_BY_ROWS", 1);
      if (sparse) {
        sk_by_rows = dev_knob("MI355_LAT_IPC", 0);
        if (!sk_by_rows) sk_by_rows = PLAN_BY_ROWS_AUTO;
        sk_slices = std::min<uint32_t>(dev_knob("MI355_LAT_SLICES_MAX", 16), std::max(1u, MERGE_PRE_SRC / nprobe));
      } else {
        sk_slices = (uint32_t)std::min<uint64_t>(dev_knob("MI355_LAT_SLICES_MAX", 8), (3ull * ix->n_cus + pairs - 1) / pairs);
      }
    }
    sk_slices = std::max(1u, std::min(sk_slices, ix->max_len / 2048u));
    if (pl.kk > 256u) sk_slices = 1;  // (multi-pass selection re-scans per pass: keep whole partitions)
    // a sliced item's `pair` word is [6 bits slices - 1][6 bits slice][20 bits pair] (sk_pack_pair): whatever the knobs and
    // the CU count say, never more slices or pairs than the fields hold
    sk_slices = std::min(sk_slices, SK_MAX_SLICES);
    if (sk_slices > 1 && pairs >= (1ull << 20)) sk_slices = 1;
    if (sk_slices == 1) sk_by_rows = 0;
  }
  const uint32_t n_slices = skew ? sk_slices : std::max(1u, (ix->max_len + slice - 1) / slice);

  // chunk the batch so the workspace stays bounded
  const size_t spill_per_item = (size_t)(ix->m - m_lds) * 1024;  // table tail of one work item (k_scan_pair SPILL)
  const size_t per_q = (size_t)ix->nlist * 4 + (size_t)nprobe * n_slices * (pl.kk * sizeof(Cand) + spill_per_item);
  const size_t budget = (size_t)(pl.ws_mb ? pl.ws_mb : dev_knob("MI355_WORKSPACE_MB", 2048)) << 20;
  uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(nq, budget / std::max<size_t>(per_q, 1)));
  chunk = std::min(chunk, 65535u);  // grid.z limit
  // Distance tables of the whole chunk by the batch-level kernels (kernels_lut.h) instead of one build per work item: where
  // the shape qualifies (dsub 16: the reference's m = dim / 16), for candidate lists the image kernels are instantiated
  // for, and not for a maximum_nprobes second pass (device-side batch size: its slots are mostly inactive).  The images
  // get their own budget (4 * M bytes per code row and pair: 48 KiB at m = 48, 16 GiB in all); if they cannot be allocated the work
  // items build their tables as before.
  bool lut_img = skew && ix->lut_img_ok && !ix->lut_inline_cfg && pl.kk <= 128u && !pl.act.n && !sk_by_rows && dev_knob("MI355_LUT_IMAGES", 1);
  if (lut_img) {
    const size_t per_q_img = (size_t)nprobe * (lut_image_bytes_per_pair(ix) + lut_residual_bytes_per_pair(ix));
    // (16 GiB: a C5 batch — 2048 queries x 64 probes x 96 KiB — must stay ONE chunk, or its deferred re-rank falls back to
    //  the serial path: a batch cut in two by an 8 GiB budget ran 35.5 ms per step where the overlapped one takes 24)
    const size_t img_budget = (size_t)dev_knob("MI355_LUT_IMAGES_MB", 16384) << 20;
    const uint32_t chunk_img = (uint32_t)std::max<size_t>(1, img_budget / std::max<size_t>(per_q_img, 1));
    const uint32_t c2 = std::min(chunk, chunk_img);
    if (ix->w_lutimg.ensure((size_t)c2 * nprobe * lut_image_bytes_per_pair(ix)) != MI355_OK ||
        ix->w_lutres.ensure((size_t)c2 * nprobe * lut_residual_bytes_per_pair(ix)) != MI355_OK) {
      (void)hipGetLastError();
      lut_img = false;  // (HBM is full: no images; the failed buffer is empty again)
    } else {
      chunk = c2;
    }
  }
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
    // latency mode, L2 / dot on the production scan: k_coarse_lat (dot chains; arms the control word) + k_select_plan
    // (scores, probe selection, work list) — two launches where k_arm_deadline, k_coarse_split, k_select_probes and
    // k_plan_sparse were four
    const bool lat_front = n == nq && lat_front_applies(ix, nq, pl);
    if (pl.arm_in_front && !(lat_front && q0 == 0) && q0 == 0) {  // (cannot happen: the caller asked the same predicate)
      hipLaunchKernelGGL(k_arm_deadline, dim3(1), dim3(1), 0, st, d_ctl, pl.arm_ticks, pl.arm_reset);
      HIP_TRY(hipGetLastError());
    }
    PlanArgs pa{};
    if (skew) {
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
      pa.by_rows = lut_img ? 0u : sk_by_rows;
      pa.n_wg = std::max(1u, scan_cus_of(ix, pl));
      pa.act = act;
    }
    if (lat_front) {
      // (the argument block always carries the 3.5 KB slot; it holds the query only for a single host query — pl.host_q)
      static thread_local CoarseLatQuery qarg_store;
      CoarseLatQuery* qarg = &qarg_store;
      if (pl.host_q) memcpy(qarg->v, pl.host_q, sizeof(float) * ix->dim);
      int lpc = 4;
      while (lpc < 16 && (uint64_t)ix->nlist * lpc < 2ull * 64 * ix->n_cus) lpc *= 2;
      if (const uint32_t f = dev_knob("MI355_LAT_LPC", 0)) lpc = (int)f;
      auto go = [&](auto kern, size_t lds, uint32_t cpw) -> int {
        if (lds > 48u * 1024) HIP_TRY(ensure_dyn_lds((const void*)kern, lds));
        hipLaunchKernelGGL(kern, dim3((ix->nlist + cpw - 1) / cpw + 1u), dim3(64), lds, st, *qarg, pl.host_q ? 1u : 0u, q, n, ix->dim, view.centroids, ix->nlist,
                           ix->w_qp.as<float>(), ix->w_qq.as<float>(), ix->w_coarse.as<float>(), pl.arm_in_front ? d_ctl : (DevCtl*)nullptr,
                           pl.arm_ticks, pl.arm_reset, ix->metric == MI355_METRIC_COSINE ? 1u : 0u);
        return MI355_OK;
      };
      const int rc = lpc == 4    ? go(k_coarse_lat<4, 24>, coarse_lat_lds<4, 24>(n, ix->dim), 16u)
                     : lpc == 8 ? go(k_coarse_lat<8, 24>, coarse_lat_lds<8, 24>(n, ix->dim), 8u)
                                : go(k_coarse_lat<16, 12>, coarse_lat_lds<16, 12>(n, ix->dim), 4u);
      if (rc != MI355_OK) return rc;
      HIP_TRY(hipGetLastError());
      if (prof) HIP_TRY(hipEventRecord(es.ev[1], st));
      SelectPlanArgs sp;
      sp.raw = ix->w_coarse.as<float>();
      sp.qq = ix->w_qq.as<float>();
      sp.cnorm = view.cnorm;
      sp.metric = ix->metric;
      sp.nlist = ix->nlist;
      sp.nprobe = nprobe;
      sp.plen = view.plen;
      sp.probes = ix->w_probes.as<uint32_t>();
      sp.stat_rows = d_stat;
      sp.qthr = ix->qthr.as<uint32_t>();
      sp.coarse_out = nullptr;
      sp.ticket = ix->heads.as<uint32_t>() + 8 * SK_HEAD_STRIDE;
      sp.plan = pa;
      sp.select_only = 0;
      const size_t sp_lds = ((size_t)ix->nlist + nprobe) * 4u;
      if (sp_lds > 40u * 1024) HIP_TRY(ensure_dyn_lds((const void*)k_select_plan, sp_lds));
      hipLaunchKernelGGL(k_select_plan, dim3(n), dim3(SELPLAN_NT), sp_lds, st, sp);
      HIP_TRY(hipGetLastError());
    } else {
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
          if (lds > 48u * 1024) HIP_TRY(ensure_dyn_lds((const void*)kern, lds));
          hipLaunchKernelGGL(kern, dim3((ix->nlist * lpc + 63) / 64), dim3(64), lds, st, q, n, ix->dim, ix->metric, view.centroids,
                             view.cnorm, ix->nlist, ix->w_qp.as<float>(), ix->w_qq.as<float>(), ix->w_coarse.as<float>());
          return MI355_OK;
        };
        const int rc = lpc == 4 ? go(k_coarse_split<4>) : lpc == 8 ? go(k_coarse_split<8>) : go(k_coarse_split<16>);
        if (rc != MI355_OK) return rc;
      } else {
      if (small_lds > 48u * 1024)
        HIP_TRY(ensure_dyn_lds((const void*)k_coarse_small, small_lds));
      hipLaunchKernelGGL(k_coarse_small, dim3((ix->nlist + 63) / 64), dim3(64), small_lds, st, q, n, ix->dim, ix->metric,
                         view.centroids, view.cnorm, ix->nlist, ix->w_qp.as<float>(), ix->w_qq.as<float>(),
                         ix->w_coarse.as<float>());
      }
    } else if (dev_knob("MI355_COARSE_VALU", 0))  // dev knob: the register-tiled VALU kernel (same bits)
      hipLaunchKernelGGL(k_coarse_tile, dim3((ix->nlist + CO_T - 1) / CO_T, (n + CO_T - 1) / CO_T),
                         dim3(256), 0, st, ix->w_qp.as<float>(), ix->w_qq.as<float>(), n,
                         view.centroids, view.cnorm, ix->nlist, ix->dim, ix->metric,
                         ix->w_coarse.as<float>());
    else if ((ix->dim & 3u) == 0 && (uint64_t)((ix->nlist + CM2_T - 1) / CM2_T) * ((n + CM2_T - 1) / CM2_T) >= 2u * (uint64_t)ix->n_cus &&
             dev_knob("MI355_COARSE_BLOCKED", 1))
      hipLaunchKernelGGL(k_coarse_mfma2, dim3((ix->nlist + CM2_T - 1) / CM2_T, (n + CM2_T - 1) / CM2_T),
                         dim3(256), 0, st, ix->w_qp.as<float>(), ix->w_qq.as<float>(), n,
                         view.centroids, view.cnorm, ix->nlist, ix->dim, ix->metric,
                         ix->w_coarse.as<float>(), act);
    else
      hipLaunchKernelGGL(k_coarse_mfma, dim3((ix->nlist + CM_T - 1) / CM_T, (n + CM_T - 1) / CM_T),
                         dim3(256), 0, st, ix->w_qp.as<float>(), ix->w_qq.as<float>(), n,
                         view.centroids, view.cnorm, ix->nlist, ix->dim, ix->metric,
                         ix->w_coarse.as<float>(), act);
    HIP_TRY(hipGetLastError());
    if (prof) HIP_TRY(hipEventRecord(es.ev[1], st));
    if (skew && ix->nlist > 8192u && ix->nlist <= SELPLAN_MAX_NLIST && nprobe <= ix->nlist && dev_knob("MI355_SELECT_WIDE", 1)) {
      // (round 6) the latency front's selection, one 1024-thread workgroup per query, for batches too: finished scores in, no plan.
      // From 8192 partitions: at 12 207 it takes 128 us per 2048 queries where k_select_probes takes 177; at 4096 it is the
      // slower one (85 against 72 us: eight 256-thread workgroups per CU hide their own latencies better than two of 1024)
      SelectPlanArgs sp{};
      sp.raw = ix->w_coarse.as<float>();
      sp.metric = ix->metric;
      sp.nlist = ix->nlist;
      sp.nprobe = nprobe;
      sp.plen = view.plen;
      sp.probes = ix->w_probes.as<uint32_t>();
      sp.stat_rows = d_stat;
      sp.qthr = ix->qthr.as<uint32_t>();
      sp.plan.act = act;
      sp.select_only = 1;
      const size_t sp_lds = ((size_t)ix->nlist + nprobe) * 4u;
      if (sp_lds > 40u * 1024) HIP_TRY(ensure_dyn_lds((const void*)k_select_plan, sp_lds));
      hipLaunchKernelGGL(k_select_plan, dim3(n), dim3(SELPLAN_NT), sp_lds, st, sp);
    } else {
    hipLaunchKernelGGL(k_select_probes, dim3(n), dim3(256), 0, st, ix->w_coarse.as<float>(),
                       ix->nlist, nprobe, view.plen, ix->w_probes.as<uint32_t>(), d_stat, act,
                       skew ? ix->qthr.as<uint32_t>() : (uint32_t*)nullptr);
    }
    HIP_TRY(hipGetLastError());
    }
    }  // !lat_front
    if (prof) HIP_TRY(hipEventRecord(es.ev[2], st));

    if (skew) {
      const uint32_t pb = (pa.n_pairs + 255) / 256;
      // (qthr, the queries' running distance bounds, was reset by k_select_probes / k_take_probes)
      if (lat_front) {
        // (k_select_plan's last workgroup wrote the work list)
      } else if (pa.n_pairs <= PLAN_SPARSE_MAX_PAIRS && dev_knob("MI355_PLAN_SPARSE", 1)) {
        hipLaunchKernelGGL(k_plan_sparse, dim3(1), dim3(PLAN_SPARSE_MAX_PAIRS), 0, st, pa);
      } else if (pa.n_pairs <= PLAN_FUSED_MAX_PAIRS && dev_knob("MI355_PLAN_FUSED", 1)) {
        hipLaunchKernelGGL(k_plan_fused, dim3(1), dim3(1024), 0, st, pa);
      } else {
        hipLaunchKernelGGL(k_plan_count, dim3(pb), dim3(256), 0, st, pa);
        hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(1024), 0, st, pa);
        hipLaunchKernelGGL(k_plan_fill, dim3(pb), dim3(256), 0, st, pa);
      }
      HIP_TRY(hipGetLastError());
      if (lut_img)
        ST_TRY(launch_lut_images(ix, ix->w_qp.as<float>(), ix->items.as<SkewItem>(), ix->q_start.as<uint32_t>(), pa.n_pairs, n_slices, nprobe,
                                 ix->w_lutres.as<float>(), ix->w_lutimg.as<float>(), st));
      if (prof) HIP_TRY(hipEventRecord(es.ev[6], st));
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
      const uint32_t scan_cus = scan_cus_of(ix, pl);
      uint32_t n_blocks = std::max(1u, (uint32_t)std::min<uint64_t>(scan_cus, (uint64_t)n * nprobe * n_slices));
      ka.n_slabs = ix->sk_slabs;
      ka.res_floats = lut_img ? 0u : ix->sk_res_floats;  // (image kernels keep no residual in LDS)
      ka.lut_img = lut_img ? ix->w_lutimg.as<float>() : nullptr;
      ka.dev_tl = nullptr;
#ifdef MI355_DEV_TIMELINE
      ST_TRY(ix->w_tl.ensure(sizeof(unsigned long long) * SK_TL_WORDS * 2u * ix->n_cus));
      HIP_TRY(hipMemsetAsync(ix->w_tl.p, 0, sizeof(unsigned long long) * SK_TL_WORDS * 2u * ix->n_cus, st));
      ka.dev_tl = ix->w_tl.as<unsigned long long>();
      ix->tl_grid = n_blocks;
#endif
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
      if (lut_img)
        ST_TRY(launch_scan_skew_img(ka, ix->sk_M, ix->sk_slabbed, n_blocks, (uint64_t)n * nprobe * n_slices, pl.kk, st));
      else if (n_slices > 1 && !ix->sk_slabbed && dev_knob("MI355_LAT_KERNEL", 1) &&
               (pl.kk <= 16u || (pl.kk <= 128u && sk_scan_lds(ix->sk_M, ix->sk_res_floats, 16, 3) <= 160u * 1024)))
        ST_TRY(launch_scan_skew_lat(ka, ix->sk_M, n_blocks, (uint64_t)n * nprobe * n_slices, pl.kk, st));
      else
        ST_TRY(launch_scan_skew(ka, ix->sk_M, ix->sk_slabbed, n_blocks, (uint64_t)n * nprobe * n_slices, pl.kk, st));
    } else {
      if (prof) HIP_TRY(hipEventRecord(es.ev[6], st));
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
                           defer ? std::max(1u, MI355_REFINE_SIDE_CUS * MI355_REFINE_SIDE_WGS_PER_CU / ((pl.kk + 255u) / 256u)) : 0u));
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
  ix->stats.lut_images = lut_img ? 1u : 0u;
  return MI355_OK;
}

