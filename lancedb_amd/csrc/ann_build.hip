// ann_build.hip — index training and population behind include/mi355_ann.h
// (SURVEY.md §8f rank 3): mi355_ivfpq_encode, mi355_kmeans_train, mi355_ivf_residuals.
// Reference: IvfBuildParams / PQBuildParams, rust/lancedb/src/index/vector.rs:61-119,
// table/create_index.rs:68-102, :283-303.
#include "ann_internal.h"
#include "kernels_ivfpq.h"
#include "kernels_encode.h"

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
                                      uint32_t* out_assign) try {
  if (!d) return fail(MI355_ERR_INVALID_INPUT, "desc is NULL");
  if (d->struct_size != sizeof(mi355_encode_desc))
    return fail(MI355_ERR_INVALID_INPUT, "mi355_encode_desc.struct_size %u != %zu (ABI mismatch)", d->struct_size,
                sizeof(mi355_encode_desc));
  if (d->dim == 0 || d->nlist == 0 || d->m == 0) return fail(MI355_ERR_INVALID_INPUT, "dim, nlist and m must be > 0");
  if (d->dim % d->m) return fail(MI355_ERR_INVALID_INPUT, "dim %u is not a multiple of m %u", d->dim, d->m);
  if (d->nbits != 8 && d->nbits != 4) return fail(MI355_ERR_INVALID_INPUT, "num_bits must be 4 or 8, got %u", d->nbits);
  if (d->nbits == 4 && d->m % 2) return fail(MI355_ERR_INVALID_INPUT, "num_sub_vectors must be even when num_bits is 4, got %u", d->m);
  if (d->metric > MI355_METRIC_DOT || d->mem > MI355_MEM_DEVICE) return fail(MI355_ERR_INVALID_INPUT, "bad metric / mem enum");
  if (!d->centroids || !d->codebook || !out_part_offsets) return fail(MI355_ERR_INVALID_INPUT, "NULL buffer");
  if (n_rows && (!vectors || !out_codes || !out_order)) return fail(MI355_ERR_INVALID_INPUT, "NULL buffer");
  const uint32_t dim = d->dim, nlist = d->nlist, m = d->m, dsub = dim / m;
  const uint32_t ks = 1u << d->nbits, mb = m * d->nbits / 8;  // codebook entries, code bytes per row
  const uint32_t jt = d->nbits == 4 ? 2 : (m % 4 == 0) ? 4 : 1;
  if ((size_t)jt * ks * dsub * 4 > 150u * 1024)
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
  ST_TRY(w.cb.ensure(sizeof(float) * (size_t)m * ks * dsub));
  ST_TRY(w.cn.ensure(sizeof(float) * nlist));
  ST_TRY(w.qp.ensure(sizeof(float) * chunk * dim));
  ST_TRY(w.qq.ensure(sizeof(float) * chunk));
  ST_TRY(w.coarse.ensure(sizeof(float) * chunk * nlist));
  ST_TRY(w.assign.ensure(sizeof(uint32_t) * n_rows));
  ST_TRY(w.hist.ensure(sizeof(uint32_t) * nlist));
  ST_TRY(w.codes_src.ensure((size_t)n_rows * mb));
  if (host) {
    ST_TRY(w.x.ensure(sizeof(float) * chunk * dim));
    ST_TRY(w.codes_dst.ensure((size_t)n_rows * mb));
    ST_TRY(w.order.ensure(sizeof(uint64_t) * n_rows));
  }
  HIP_TRY(copy_in(w.cen.p, d->centroids, sizeof(float) * (size_t)nlist * dim, d->mem, st));
  HIP_TRY(copy_in(w.cb.p, d->codebook, sizeof(float) * (size_t)m * ks * dsub, d->mem, st));
  hipLaunchKernelGGL(k_centroid_norms, dim3((nlist + 63) / 64), dim3(64), 0, st, w.cen.as<float>(), nlist, dim,
                     w.cn.as<float>());
  HIP_TRY(hipMemsetAsync(w.hist.p, 0, sizeof(uint32_t) * nlist, st));

  // ---- pass A: partition + codes of every row, in source order
  const size_t enc_lds = (size_t)jt * ks * dsub * 4;
  if (d->nbits == 4)
    HIP_TRY(hipFuncSetAttribute((const void*)k_encode_rows<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_lds));
  else if (jt == 4)
    HIP_TRY(hipFuncSetAttribute((const void*)k_encode_rows<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_lds));
  else
    HIP_TRY(hipFuncSetAttribute((const void*)k_encode_rows<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)enc_lds));
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
                  dim, m, dsub, d->metric, mb, w.codes_src.as<uint8_t>()};
    if (d->nbits == 4)
      hipLaunchKernelGGL((k_encode_rows<2, 4>), dim3((n + 255) / 256, m / 2), dim3(256), enc_lds, st, ea);
    else if (jt == 4)
      hipLaunchKernelGGL((k_encode_rows<4, 8>), dim3((n + 255) / 256, m / 4), dim3(256), enc_lds, st, ea);
    else
      hipLaunchKernelGGL((k_encode_rows<1, 8>), dim3((n + 255) / 256, m), dim3(256), enc_lds, st, ea);
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
    const uint32_t pw = (mb % 4 == 0) ? 4 : 1;  // bytes per thread
    const uint64_t items = n_rows * (mb / pw);
    const uint64_t blocks = (items + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return fail(MI355_ERR_NOT_SUPPORTED, "too many rows for one permute launch");
    if (pw == 4)
      hipLaunchKernelGGL(k_permute_codes<4>, dim3((uint32_t)blocks), dim3(256), 0, st, w.codes_src.as<uint8_t>(),
                         d_order, n_rows, mb, d_codes);
    else
      hipLaunchKernelGGL(k_permute_codes<1>, dim3((uint32_t)blocks), dim3(256), 0, st, w.codes_src.as<uint8_t>(),
                         d_order, n_rows, mb, d_codes);
    HIP_TRY(hipGetLastError());
  }
  if (host) {
    HIP_TRY(hipMemcpyAsync(out_codes, d_codes, (size_t)n_rows * mb, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(out_order, d_order, sizeof(uint64_t) * n_rows, hipMemcpyDeviceToHost, st));
  }
  if (out_assign)
    HIP_TRY(hipMemcpyAsync(out_assign, w.assign.p, sizeof(uint32_t) * n_rows,
                           host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));
  return MI355_OK;
} MI355_ABI_GUARD("mi355_ivfpq_encode")

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

// Lloyd iterations on scratch `w` (its stream): rows at `vectors` (memory per d->mem, stride d->ld),
// centroids already in w.cen; leaves the trained centroids in w.cen and the last histogram in `hist`
static int32_t lloyd_on_scratch(TrainScratch& w, const mi355_kmeans_desc* d, const float* vectors, uint64_t n_rows,
                                std::vector<uint32_t>& hist) {
  hipStream_t st = w.st;
  const uint32_t dim = d->dim, k = d->k;
  hist.assign(k, 0);
  if (!n_rows || !d->iters) return MI355_OK;
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
  return MI355_OK;
}

extern "C" int32_t mi355_kmeans_train(const mi355_kmeans_desc* d, const float* vectors, uint64_t n_rows,
                                      float* centroids, uint64_t* out_counts) try {
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
  std::vector<uint32_t> hist;
  ST_TRY(lloyd_on_scratch(w, d, vectors, n_rows, hist));
  if (n_rows && d->iters)
    HIP_TRY(hipMemcpyAsync(centroids, w.cen.p, sizeof(float) * (size_t)k * dim,
                           host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
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
} MI355_ABI_GUARD("mi355_kmeans_train")

// All m PQ sub-quantisers in one call: the residual matrix crosses to the device once and the m
// trainers share one scratch set and one stream (sub-quantiser j = mi355_kmeans_train on columns
// [j * dsub, (j + 1) * dsub) with ld = dim, bit for bit).
extern "C" int32_t mi355_pq_train(const mi355_pq_train_desc* d, const float* residuals, uint64_t n_rows, float* codebook) try {
  if (!d) return fail(MI355_ERR_INVALID_INPUT, "desc is NULL");
  if (d->struct_size != sizeof(mi355_pq_train_desc))
    return fail(MI355_ERR_INVALID_INPUT, "mi355_pq_train_desc.struct_size %u != %zu (ABI mismatch)", d->struct_size,
                sizeof(mi355_pq_train_desc));
  if (d->dim == 0 || d->m == 0 || d->dim % d->m) return fail(MI355_ERR_INVALID_INPUT, "dim %u is not a multiple of m %u", d->dim, d->m);
  if (d->nbits != 8 && d->nbits != 4) return fail(MI355_ERR_INVALID_INPUT, "num_bits must be 4 or 8, got %u", d->nbits);
  if (d->metric > MI355_METRIC_DOT || d->mem > MI355_MEM_DEVICE) return fail(MI355_ERR_INVALID_INPUT, "bad metric / mem enum");
  if (!codebook || (n_rows && !residuals)) return fail(MI355_ERR_INVALID_INPUT, "NULL buffer");
  if (n_rows >> 40) return fail(MI355_ERR_NOT_SUPPORTED, "too many training rows");
  ST_TRY(need_device(d->device));
  const uint32_t dim = d->dim, m = d->m, dsub = dim / m, ks = 1u << d->nbits;
  const bool host = d->mem == MI355_MEM_HOST;
  TrainScratch w;
  HIP_TRY(hipStreamCreateWithFlags(&w.st, hipStreamNonBlocking));
  hipStream_t st = w.st;
  ScratchBuf d_res, d_cb;
  const float* res = residuals;
  if (host && n_rows) {
    ST_TRY(d_res.ensure(sizeof(float) * n_rows * dim));
    HIP_TRY(hipMemcpyAsync(d_res.p, residuals, sizeof(float) * n_rows * dim, hipMemcpyHostToDevice, st));
    res = d_res.as<float>();
  }
  ST_TRY(d_cb.ensure(sizeof(float) * (size_t)m * ks * dsub));
  HIP_TRY(copy_in(d_cb.p, codebook, sizeof(float) * (size_t)m * ks * dsub, d->mem, st));
  mi355_kmeans_desc kd{};
  kd.struct_size = sizeof(kd);
  kd.dim = dsub;
  kd.k = ks;
  // residuals are already centred / normalised: plain L2 k-means, or dot for dot indexes
  kd.metric = d->metric == MI355_METRIC_DOT ? MI355_METRIC_DOT : MI355_METRIC_L2;
  kd.iters = d->iters;
  kd.mem = MI355_MEM_DEVICE;
  kd.device = d->device;
  kd.ld = dim;
  ST_TRY(check_kmeans_desc(&kd));
  ST_TRY(w.cen.ensure(sizeof(float) * (size_t)ks * dsub));
  std::vector<uint32_t> hist;
  for (uint32_t j = 0; j < m; ++j) {
    float* cbj = d_cb.as<float>() + (size_t)j * ks * dsub;
    HIP_TRY(hipMemcpyAsync(w.cen.p, cbj, sizeof(float) * (size_t)ks * dsub, hipMemcpyDeviceToDevice, st));
    ST_TRY(lloyd_on_scratch(w, &kd, res + (size_t)j * dsub, n_rows, hist));
    HIP_TRY(hipMemcpyAsync(cbj, w.cen.p, sizeof(float) * (size_t)ks * dsub, hipMemcpyDeviceToDevice, st));
  }
  HIP_TRY(hipMemcpyAsync(codebook, d_cb.p, sizeof(float) * (size_t)m * ks * dsub,
                         host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));
  return MI355_OK;
} MI355_ABI_GUARD("mi355_pq_train")

extern "C" int32_t mi355_ivf_residuals(const mi355_kmeans_desc* d, const float* vectors, uint64_t n_rows,
                                       const float* centroids, float* out_residuals, uint32_t* out_assign) try {
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
} MI355_ABI_GUARD("mi355_ivf_residuals")
