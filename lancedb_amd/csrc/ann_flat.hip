// ann_flat.hip — the flat (no index / bypass_vector_index) handle behind
// include/mi355_ann.h: bf16 MFMA GEMM filter + exact re-rank, or the exact sweep.
// Replaces KNNVectorDistance + SortExec TopK
// (/root/reference/python/python/lancedb/query.py:1365-1370, rust/lancedb/src/query.rs:1360-1370).
#include "ann_internal.h"
#include "kernels_flat.h"
#include "kernels_flat_mfma.h"
#include "kernels_flat_mfma8.h"

// what MI355_FLAT_GEMM_AUTO means for batches > 128 queries: the variant validated and measured
// fastest on hardware (profiles/r02_e_flat_gemm_schedules.json: 8-phase, one workgroup per tile,
// 14.45 ms against 15.46 ms for the two-barrier kernel at 10 M x 768); flipped only together with
// a committed A/B
#define MI355_FLAT_GEMM_AUTO_BIG MI355_FLAT_GEMM_8PHASE

// -------------------------------------------------------------------- flat --
extern "C" int32_t mi355_flat_open(const mi355_flat_desc* d, mi355_flat** out) try {
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
    const bool force_exact = dev_knob("MI355_FLAT_EXACT", 0) != 0;  // dev knob: keep the scalar sweep only
    if (!force_exact && d->n_rows >= dev_knob("MI355_FLAT_MFMA_MIN_ROWS", 4096)) {
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
} MI355_ABI_GUARD("mi355_flat_open")

// The MFMA filter + exact re-rank over queries [d_q, d_q + n) (device), results in d_ids/d_dist/d_cnt
static int32_t run_flat_mfma(mi355_flat* f, const float* d_q, uint32_t nq, uint32_t metric, uint32_t k,
                             const RangeFilter& range, uint64_t* d_ids, float* d_dist, uint32_t* d_cnt) {
  hipStream_t st = f->stream;
  // GEMM schedule (mi355_flat_configure): 128 x 128 (4 waves, 2 workgroups per CU) for small batches,
  // otherwise a 256 x 256 tile: the 8-phase schedule when the k loop has >= 2 tiles
  uint32_t variant = f->gemm_variant;
  const uint32_t KT = f->dimp / FG_BK;
  if (variant == MI355_FLAT_GEMM_AUTO) variant = nq <= 128 ? MI355_FLAT_GEMM_128 : MI355_FLAT_GEMM_AUTO_BIG;
  if (variant >= MI355_FLAT_GEMM_8PHASE && KT < 2)
    variant = MI355_FLAT_GEMM_256;  // the 8-phase walk stages two k-tiles ahead
  const bool oct = variant >= MI355_FLAT_GEMM_8PHASE;
  const bool big = variant == MI355_FLAT_GEMM_256 || oct;
  const uint32_t BM = big ? 256 : 128, BN = BM;
  const uint32_t n_rtiles = (uint32_t)((f->n_rows + BM - 1) / BM);
  const uint32_t n_groups = n_rtiles * (BM / FG_GROUP);
  uint32_t groups_per_seg = (n_groups + FG_MAX_SEG - 1) / FG_MAX_SEG;
  const uint32_t n_seg = (n_groups + groups_per_seg - 1) / groups_per_seg;
  // bound the group-minimum matrix (n_groups x queries f32) to ~2 GiB per pass
  const size_t budget = (size_t)dev_knob("MI355_WORKSPACE_MB", 2048) << 20;
  uint32_t chunk = (uint32_t)std::min<size_t>(((size_t)nq + BN - 1) / BN * BN,
                                              std::max<size_t>(BN, (budget / ((size_t)n_groups * 4)) / BN * BN));
  const int kpl = kpl_for(k);
  const bool want_sum = (f->cfg_flags & MI355_FLAT_CHECKSUM) != 0;
  // MI355_FLAT_CHECKSUM also synchronises after every stage, which localises a device fault
  auto stage_ok = [&](const char* what) -> int32_t {
    if (!want_sum) return MI355_OK;
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(MI355_ERR_RUNTIME, "flat stage %s failed: %s", what, hipGetErrorString(e));
    return MI355_OK;
  };
  if (want_sum) {
    f->checksum = 0;
    f->census[0] = f->census[1] = 0;
    f->census_sum = 0.0;
  }
  ST_TRY(f->w_fallback.ensure(16));
  HIP_TRY(hipMemsetAsync(f->w_fallback.p, 0, 4, st));
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
    const bool prof = (f->cfg_flags & MI355_FLAT_PROFILE) != 0;
    mi355_flat::FlatEv fe{};
    if (prof) {
      if (!f->ev_free.empty()) {
        fe = f->ev_free.back();
        f->ev_free.pop_back();
      } else {
        for (auto& e : fe.ev) HIP_TRY(hipEventCreate(&e));
      }
      HIP_TRY(hipEventRecord(fe.ev[0], st));
    }
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
    ga.vw = nullptr;
    if (variant == MI355_FLAT_GEMM_8PHASE && metric != MI355_METRIC_L2) {
      DevBuf& vw = metric == MI355_METRIC_COSINE ? f->vw_cos : f->vw_dot;
      if (!vw.p) {  // once per column and metric
        const uint64_t vv_rows = ((f->n_rows + 255) / 256) * 256;
        ST_TRY(vw.ensure(sizeof(float) * vv_rows));
        hipLaunchKernelGGL(k_flat_row_factor, dim3((uint32_t)((vv_rows + 255) / 256)), dim3(256), 0, st, f->vv.as<float>(), vv_rows,
                           metric, vw.as<float>());
        HIP_TRY(hipGetLastError());
      }
      ga.vw = vw.as<float>();
    }
    uint32_t gemm_blocks = ((n_rtiles + 7) / 8) * 8 * ga.n_qtiles;  // one per (row tile, query tile)
    // persistent grid: one workgroup per CU slot walks its XCD's tiles and overlaps the next tile's
    // first stage with the current tile's last k-step and epilogue (grid_workgroups of
    // mi355_flat_configure: 1 = one workgroup per tile, N >= 8 = a persistent grid of N / 8 * 8,
    // 0 = what measured fastest for the schedule: persistent for the two-barrier kernels, one
    // workgroup per tile for the 8-phase one, whose staggered wave groups already hide the prologue)
    if (f->grid_workgroups != 1 && !(oct && f->grid_workgroups == 0)) {
      uint32_t slots = f->grid_workgroups / 8 * 8;
      if (f->grid_workgroups < 8) {
        int cus = 0;
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, f->device));
        slots = (uint32_t)std::max(cus, 8) / 8 * 8 * (big ? 1u : 2u);
      }
      gemm_blocks = std::min(gemm_blocks, slots);
    }
    // the 8-phase kernel keeps the epilogue's operands (|v|^2, qa, qg of the tile: 3 KiB) behind the stage buffers
    const size_t gemm_lds = (size_t)2 * (BM + BN) * FG_BK * 2 + (oct ? 3072u : 0u);
#define LAUNCH_FG(MET)                                                                              \
  {                                                                                                 \
    if (oct) {                                                                                      \
      auto kern = variant == MI355_FLAT_GEMM_8PHASE ? k_flat_gemm8<MET, 1> : k_flat_gemm8<MET, 0>;  \
      HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                  (int)gemm_lds));                                                  \
      hipLaunchKernelGGL(kern, dim3(gemm_blocks), dim3(512), gemm_lds, st, ga);                     \
    } else if (big) {                                                                               \
      auto kern = k_flat_gemm<MET, 2, 4, 8, 4>;                                                     \
      HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                  (int)gemm_lds));                                                  \
      hipLaunchKernelGGL(kern, dim3(gemm_blocks), dim3(512), gemm_lds, st, ga);                     \
    } else {                                                                                        \
      auto kern = k_flat_gemm<MET, 2, 2, 4, 4>;                                                     \
      HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                  (int)gemm_lds));                                                  \
      hipLaunchKernelGGL(kern, dim3(gemm_blocks), dim3(256), gemm_lds, st, ga);                     \
    }                                                                                               \
  }
    if (prof) HIP_TRY(hipEventRecord(fe.ev[1], st));
    if (metric == MI355_METRIC_L2) LAUNCH_FG(MI355_METRIC_L2)
    else if (metric == MI355_METRIC_COSINE) LAUNCH_FG(MI355_METRIC_COSINE)
    else LAUNCH_FG(MI355_METRIC_DOT)
#undef LAUNCH_FG
    HIP_TRY(hipGetLastError());
    if (prof) {
      HIP_TRY(hipEventRecord(fe.ev[2], st));
      f->fstats.gemm_flops += 2ull * n_pad * (uint64_t)f->n_rows * f->dimp;
    }
    f->fstats.gemm_variant = variant;
    ST_TRY(stage_ok("gemm"));
    if (want_sum) {
      // only whole tiles' groups of real rows are defined; padding queries are computed too
      ST_TRY(f->w_sum.ensure(32));
      unsigned long long h_sum[4] = {0, 0, 0, 0};
      HIP_TRY(hipMemsetAsync(f->w_sum.p, 0, 32, st));
      const size_t real_groups = (size_t)((f->n_rows + FG_GROUP - 1) / FG_GROUP);
      hipLaunchKernelGGL(k_flat_checksum, dim3(1024), dim3(256), 0, st, ga.gm, real_groups * n_pad,
                         f->w_sum.as<unsigned long long>());
      HIP_TRY(hipMemcpyAsync(h_sum, f->w_sum.p, 32, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      f->checksum += h_sum[0];
      f->census[0] += h_sum[1];
      f->census[1] += h_sum[2];
      double fs;
      memcpy(&fs, &h_sum[3], 8);
      f->census_sum += fs;
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
    ra.fallback = f->w_fallback.as<uint32_t>();
    const size_t rl = (((size_t)f->dim * 4 + 15) & ~(size_t)15) + sizeof(Cand) * 4 * std::min<uint32_t>(k, 64u * kpl);
    launch_by_kpl(kpl, k_flat_rerank<1>, k_flat_rerank<2>, k_flat_rerank<4>, dim3(n), dim3(256), rl, st, ra);
    HIP_TRY(hipGetLastError());
    ST_TRY(stage_ok("rerank"));
    if (prof) {
      HIP_TRY(hipEventRecord(fe.ev[3], st));
      f->ev_pending.push_back(fe);
    }
  }
  return MI355_OK;
}

extern "C" int32_t mi355_flat_close(mi355_flat* f) try {
  if (!f) return MI355_OK;
  (void)hipSetDevice(f->device);
  DevBuf* bufs[] = {&f->vectors, &f->row_ids, &f->w_q,  &f->w_cand, &f->w_ids,  &f->w_dist, &f->w_cnt,
                    &f->shadow,  &f->vv,      &f->vw_cos,  &f->vw_dot, &f->vmax, &f->g_qb,   &f->g_qa,   &f->g_qg,   &f->g_slack,
                    &f->g_tau,   &f->g_gm,    &f->g_seg, &f->g_cnt, &f->g_cand, &f->w_filter, &f->w_sum, &f->w_fallback};
  for (DevBuf* b : bufs) b->release();
  if (f->h_pin) (void)hipHostFree(f->h_pin);
  for (auto* v : {&f->ev_free, &f->ev_pending})
    for (auto& fe : *v)
      for (auto& e : fe.ev) (void)hipEventDestroy(e);
  if (f->own_stream) (void)hipStreamDestroy(f->own_stream);
  delete f;
  return MI355_OK;
} MI355_ABI_GUARD("mi355_flat_close")

extern "C" int32_t mi355_flat_set_stream(mi355_flat* f, void* hip_stream) try {
  if (!f) return fail(MI355_ERR_INVALID_INPUT, "flat handle is NULL");
  std::lock_guard<std::mutex> lk(f->mu);
  f->stream = hip_stream ? (hipStream_t)hip_stream : f->own_stream;
  return MI355_OK;
} MI355_ABI_GUARD("mi355_flat_set_stream")

extern "C" int32_t mi355_flat_sync(mi355_flat* f) try {
  if (!f) return fail(MI355_ERR_INVALID_INPUT, "flat handle is NULL");
  HIP_TRY(hipSetDevice(f->device));
  HIP_TRY(hipStreamSynchronize(f->stream));
  return MI355_OK;
} MI355_ABI_GUARD("mi355_flat_sync")

// the device work of a flat search over device-resident queries (f->mu held by the caller; stream work only)
int32_t run_flat_search_device(mi355_flat* f, const float* d_q, uint32_t nq, const mi355_search_params* p,
                               uint64_t* d_ids, float* d_dist, uint32_t* d_cnt) {
  hipStream_t st = f->stream;
  const uint32_t k = p->k;
  const uint32_t metric = p->metric == MI355_METRIC_DEFAULT ? (uint32_t)MI355_METRIC_L2 : p->metric;
  const int kpl = kpl_for(k);  // k > 256: the selection kernels run in passes of 256 rows
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
  bool use_mfma = f->mfma && !p->has_lower_bound && flt.mode == MI355_FILTER_NONE;
  if (use_mfma && (f->cfg_flags & MI355_FLAT_FORCE_SWEEP)) use_mfma = false;
  if (use_mfma && !(f->cfg_flags & MI355_FLAT_FORCE_FILTER) && f->gemm_variant == MI355_FLAT_GEMM_AUTO) {
    // Both paths are exact and return the same bits; take the cheaper one for THIS call (measured on MI355X,
    // tests/tools/flat_path_time.py): the sweep reads the column once per query at ~2.5 TB/s
    // behind ~30 us of launches; the filter runs whole 256-query GEMM tiles at ~1 PFLOP/s behind ~0.3 ms of launches
    // (query prep, GEMM, segment minima, thresholds, compaction, re-rank) and re-ranks a few hundred rows per query.
    const double col_bytes = (double)f->n_rows * f->dim * (f->dtype == MI355_DTYPE_F32 ? 4.0 : 2.0);
    const double sweep_us = 30.0 + (double)nq * col_bytes / 2.5e6;
    const double tiles = (double)((nq + 255u) / 256u);
    const double filter_us = 300.0 + tiles * (double)f->n_rows * 256.0 * (double)f->dimp * 2.0 / 1.0e9;
    use_mfma = filter_us < sweep_us;
  }
  f->last_path = use_mfma ? 1 : 2;
  if (use_mfma) {
    ST_TRY(run_flat_mfma(f, d_q, nq, metric, k, rng, d_ids, d_dist, d_cnt));
  } else {
  // enough work items to fill 256 CUs; at least one row per thread of a 256-thread block (a table of 100 k rows is
  // 391 blocks of one row per thread: the call's time is then one row's chain, not four)
  uint32_t slice = (uint32_t)std::max<uint64_t>(256, (f->n_rows + 2047) / 2048);
  slice = (slice + 255u) & ~255u;
  const uint32_t n_slices = (uint32_t)std::max<uint64_t>(1, (f->n_rows + slice - 1) / slice);
  // the per-slice candidate slots stay within ~2 GiB
  const uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(std::min(nq, 65535u), ((size_t)2048 << 20) / ((size_t)n_slices * k * sizeof(Cand))));
  ST_TRY(f->w_cand.ensure(sizeof(Cand) * (size_t)chunk * n_slices * k));
  for (uint32_t q0 = 0; q0 < nq; q0 += chunk) {
    const uint32_t n = std::min(chunk, nq - q0);
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
    size_t lds = (((size_t)f->dim * 4 + 15) & ~(size_t)15) + sizeof(Cand) * 4 * std::min<uint32_t>(k, 64u * kpl);
    launch_by_kpl(kpl, k_flat_scan<1>, k_flat_scan<2>, k_flat_scan<4>, dim3(n_slices, 1, n), dim3(256), lds, st, fa);
    HIP_TRY(hipGetLastError());
    MergeArgs ma = merge_args_dense(f->w_cand.as<Cand>(), n_slices, k, n, k);
    ma.out_ids = d_ids + (size_t)q0 * k;
    ma.out_dist = d_dist + (size_t)q0 * k;
    ma.out_cnt = d_cnt + q0;
    launch_by_kpl(kpl, k_merge_cands<1>, k_merge_cands<2>, k_merge_cands<4>, dim3(n), dim3(64), 0, st, ma);
    HIP_TRY(hipGetLastError());
  }
  }  // exact sweep
  return MI355_OK;
}

extern "C" int32_t mi355_flat_search(mi355_flat* f, const float* queries, uint32_t n_queries,
                                     const mi355_search_params* p, uint64_t* out_rowids,
                                     float* out_dist, uint32_t* out_counts) try {
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
  auto t_start = std::chrono::steady_clock::now();
  const float* d_q = queries;
  uint64_t* d_ids = out_rowids;
  float* d_dist = out_dist;
  uint32_t* d_cnt = out_counts;
  // Small host batches (single queries: BASELINE configs[0]) travel through ONE page-locked block, as on the IVF-PQ
  // handle: pageable copies stage (and, device-to-host, block) per call — four round trips for one query's results.
  const size_t q_bytes = sizeof(float) * (size_t)n_queries * f->dim;
  const size_t q_pad = (q_bytes + 63) & ~(size_t)63;
  const size_t r_bytes = (size_t)n_queries * k * (sizeof(uint64_t) + sizeof(float)) + sizeof(uint32_t) * (size_t)n_queries;
  const bool pinned = host_io && q_pad + r_bytes <= ((size_t)4 << 20);
  if (host_io) {
    ST_TRY(f->w_q.ensure(q_bytes));
    if (pinned) {
      const size_t need = q_pad + r_bytes + 64;
      if (f->h_pin_cap < need) {
        if (f->h_pin) (void)hipHostFree(f->h_pin);
        f->h_pin = nullptr;
        f->h_pin_cap = 0;
        HIP_TRY(hipHostMalloc(&f->h_pin, need * 2, hipHostMallocDefault));
        f->h_pin_cap = need * 2;
      }
      memcpy(f->h_pin, queries, q_bytes);
      HIP_TRY(hipMemcpyAsync(f->w_q.p, f->h_pin, q_bytes, hipMemcpyHostToDevice, st));
      ST_TRY(f->w_ids.ensure(r_bytes));  // ids | distances | counts carved out of one buffer: one copy back
      d_ids = f->w_ids.as<uint64_t>();
      d_dist = (float*)(d_ids + (size_t)n_queries * k);
      d_cnt = (uint32_t*)(d_dist + (size_t)n_queries * k);
    } else {
      ST_TRY(f->w_ids.ensure(sizeof(uint64_t) * (size_t)n_queries * k));
      ST_TRY(f->w_dist.ensure(sizeof(float) * (size_t)n_queries * k));
      ST_TRY(f->w_cnt.ensure(sizeof(uint32_t) * n_queries));
      HIP_TRY(hipMemcpyAsync(f->w_q.p, queries, q_bytes, hipMemcpyHostToDevice, st));
      d_ids = f->w_ids.as<uint64_t>();
      d_dist = f->w_dist.as<float>();
      d_cnt = f->w_cnt.as<uint32_t>();
    }
    d_q = f->w_q.as<float>();
  }
  ST_TRY(run_flat_search_device(f, d_q, n_queries, p, d_ids, d_dist, d_cnt));
  if (host_io) {
    if (pinned) {
      unsigned char* h_res = (unsigned char*)f->h_pin + q_pad;
      HIP_TRY(hipMemcpyAsync(h_res, d_ids, r_bytes, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      const uint64_t* r_ids = (const uint64_t*)h_res;
      const float* r_dist = (const float*)(r_ids + (size_t)n_queries * k);
      const uint32_t* r_cnt = (const uint32_t*)(r_dist + (size_t)n_queries * k);
      memcpy(out_rowids, r_ids, sizeof(uint64_t) * (size_t)n_queries * k);
      memcpy(out_dist, r_dist, sizeof(float) * (size_t)n_queries * k);
      memcpy(out_counts, r_cnt, sizeof(uint32_t) * n_queries);
    } else {
      HIP_TRY(hipMemcpyAsync(out_rowids, d_ids, sizeof(uint64_t) * (size_t)n_queries * k, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(out_dist, d_dist, sizeof(float) * (size_t)n_queries * k, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(out_counts, d_cnt, sizeof(uint32_t) * n_queries, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
    }
    if (p->timeout_ms) {
      auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_start).count();
      if (ms > (long long)p->timeout_ms)
        return fail(MI355_ERR_TIMEOUT, "Query timeout: %lld ms > %u ms", (long long)ms, p->timeout_ms);
    }
  }
  return MI355_OK;
} MI355_ABI_GUARD("mi355_flat_search")

extern "C" int32_t mi355_flat_configure(mi355_flat* f, uint32_t gemm_variant, uint32_t grid_workgroups,
                                        uint32_t flags) try {
  if (!f) return fail(MI355_ERR_INVALID_INPUT, "flat handle is NULL");
  if (gemm_variant > MI355_FLAT_GEMM_8PHASE_REF || gemm_variant == 3)
    return fail(MI355_ERR_INVALID_INPUT, "unknown gemm variant %u", gemm_variant);
  if ((flags & MI355_FLAT_FORCE_FILTER) && (flags & MI355_FLAT_FORCE_SWEEP))
    return fail(MI355_ERR_INVALID_INPUT, "MI355_FLAT_FORCE_FILTER and MI355_FLAT_FORCE_SWEEP exclude each other");
  if (flags & ~(uint32_t)(MI355_FLAT_CHECKSUM | MI355_FLAT_PROFILE | MI355_FLAT_FORCE_FILTER | MI355_FLAT_FORCE_SWEEP)) return fail(MI355_ERR_INVALID_INPUT, "unknown flags 0x%x", flags);
  std::lock_guard<std::mutex> lk(f->mu);
  f->gemm_variant = gemm_variant;
  f->grid_workgroups = grid_workgroups;
  f->cfg_flags = flags;
  HIP_TRY(hipSetDevice(f->device));
  HIP_TRY(hipStreamSynchronize(f->stream));
  for (auto& fe : f->ev_pending) f->ev_free.push_back(fe);
  f->ev_pending.clear();
  f->fstats = mi355_flat_stats{};
  return MI355_OK;
} MI355_ABI_GUARD("mi355_flat_configure")

extern "C" int32_t mi355_flat_last_stats(mi355_flat* f, mi355_flat_stats* out) try {
  if (!f || !out) return fail(MI355_ERR_INVALID_INPUT, "NULL argument");
  if (out->struct_size != sizeof(mi355_flat_stats)) return fail(MI355_ERR_INVALID_INPUT, "mi355_flat_stats.struct_size mismatch");
  std::lock_guard<std::mutex> lk(f->mu);
  HIP_TRY(hipSetDevice(f->device));
  HIP_TRY(hipStreamSynchronize(f->stream));
  for (auto& fe : f->ev_pending) {
    float ms[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i) HIP_TRY(hipEventElapsedTime(&ms[i], fe.ev[i], fe.ev[i + 1]));
    f->fstats.us_gemm += ms[1] * 1000.f;
    f->fstats.us_rest += (ms[0] + ms[2]) * 1000.f;
    f->fstats.gemm_launches += 1;
    f->ev_free.push_back(fe);
  }
  f->ev_pending.clear();
  f->fstats.fallback_queries = 0;
  if (f->w_fallback.p) HIP_TRY(hipMemcpy(&f->fstats.fallback_queries, f->w_fallback.p, 4, hipMemcpyDeviceToHost));
  *out = f->fstats;
  out->struct_size = sizeof(mi355_flat_stats);
  return MI355_OK;
} MI355_ABI_GUARD("mi355_flat_last_stats")

extern "C" int32_t mi355_flat_checksum(mi355_flat* f, uint64_t* out) try {
  if (!f || !out) return fail(MI355_ERR_INVALID_INPUT, "NULL argument");
  std::lock_guard<std::mutex> lk(f->mu);
  *out = f->checksum;
  return MI355_OK;
} MI355_ABI_GUARD("mi355_flat_checksum")

extern "C" int32_t mi355_flat_census(mi355_flat* f, uint64_t* out_never_filter, uint64_t* out_not_finite, double* out_sum) try {
  if (!f) return fail(MI355_ERR_INVALID_INPUT, "flat handle is NULL");
  std::lock_guard<std::mutex> lk(f->mu);
  if (out_never_filter) *out_never_filter = f->census[0];
  if (out_not_finite) *out_not_finite = f->census[1];
  if (out_sum) *out_sum = f->census_sum;
  return MI355_OK;
} MI355_ABI_GUARD("mi355_flat_census")

extern "C" int32_t mi355_flat_info(const mi355_flat* f, uint32_t* out_last_path, uint32_t* out_has_filter) try {
  if (!f) return fail(MI355_ERR_INVALID_INPUT, "flat handle is NULL");
  if (out_last_path) *out_last_path = f->last_path;
  if (out_has_filter) *out_has_filter = f->mfma ? 1u : 0u;
  return MI355_OK;
} MI355_ABI_GUARD("mi355_flat_info")
