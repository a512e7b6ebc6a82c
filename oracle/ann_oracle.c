/*
 * ann_oracle.c — CPU restatement of LanceDB's vector-search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product
 * (lancedb_amd/, include/mi355_ann.h) never calls into it.
 *
 * What is restated.  The reference (lancedb 0.38.0-beta.4) delegates all
 * arithmetic on this path to the un-vendored crates lance / lance-index /
 * lance-linalg = 11.0.0-beta.19 (git tag v11.0.0-beta.19, rev
 * 3128c0024427cb5bf8c04d492893ae45e78b0511; /root/reference/Cargo.toml:16-29,
 * Cargo.lock:4817-4819).  Their source is absent, so the IVF-PQ algorithm is
 * restated from its published design and from the in-tree call sites:
 *   - request semantics: rust/lancedb/src/query.rs:1066-1114 (defaults),
 *     :1232-1288 (nprobes, distance_range [lower, upper)), :1302-1332 (refine:
 *     take limit*refine_factor by approximate distance, re-rank by true
 *     distance), :1360-1370 (bypass index = flat search)
 *   - k = limit + offset: rust/lancedb/src/table/query.rs:231
 *   - metrics: rust/lancedb/src/lib.rs:236-260
 *   - result order (_distance ASC, _rowid ASC), NULL distances dropped:
 *     python/python/lancedb/query.py:1365-1370
 *   - index shape (nlist, m, 8 or 4 bits; 4 bits need an even m):
 *     rust/lancedb/src/index/vector.rs:266-319, table/create_index.rs:96-101
 * [EXT] (lance-index, restated, not verifiable here): PQ of the residual
 * q - centroid for L2/cosine, no residual for dot; distance table
 * LUT[j][c] = l2(r_j, codebook[j][c]) (or 1 - dot); ADC
 * dist = sum_j LUT[j][code[j]] accumulated j = 0..m-1 with plain f32 adds over
 * per-partition transposed codes; cosine = L2 over unit vectors / 2;
 * dot = sum - (m - 1).
 *
 * PARITY STATUS.  Flat L2 / cosine / dot are pinned against the reference's
 * own test expectations (tests/golden/flat_reference_cases.json, transcribed
 * from python/python/tests/test_query.py and test_db.py).  IVF-PQ: PARITY
 * UNPINNED — no in-tree test fixes a codebook, a LUT value or an ADC result
 * (python/python/tests/test_query.py:1095-1097 says so), so this file is the
 * definition of correct IVF-PQ results for this repository.
 *
 * ARITHMETIC CONTRACT (what the HIP kernels reproduce bit for bit)
 *   chain_dot(a,b,n)  : acc=0; for d=0..n-1: acc = fmaf(a[d], b[d], acc)
 *   chain_l2(a,b,n)   : acc=0; for d=0..n-1: t = a[d]-b[d]; acc = fmaf(t,t,acc)
 *   cosine query      : q[d] / sqrtf(chain_dot(q,q))  (IEEE div and sqrt)
 *   coarse L2/cosine  : fmaf(-2, chain_dot(q,c), chain_dot(q,q)+chain_dot(c,c))
 *   coarse dot        : 1 - chain_dot(q,c)
 *   probes            : the nprobe smallest by (coarse, partition id)
 *   residual          : r[d] = q[d] - c_p[d]
 *   LUT[j][c]         : chain_l2(r_j, cb_jc, dsub)   | dot: 1 - chain_dot(q_j, cb_jc)
 *   ADC               : acc = 0; for j=0..m-1: acc = acc + LUT[j][code_j]
 *   4-bit codes       : LUT[j] has 16 entries; byte t of a row packs code_{2t} (low nibble) and
 *                       code_{2t+1} (high nibble); the adds stay in j order ([EXT] packing)
 *   final             : L2 acc | cosine acc*0.5f | dot acc - (float)(m-1)
 *   exact (flat/refine): L2 chain_l2(q,v) | cosine 1 - dot/(sqrtf(qq)*sqrtf(vv))
 *                        | dot 1 - chain_dot(q,v)
 * Compile with -ffp-contract=off so that only the explicit fmaf()s fuse.
 */
#define _GNU_SOURCE
#include "../include/mi355_ann.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#ifdef __AVX2__
#include <immintrin.h>
#endif

typedef struct orc_index {
  uint32_t dim, nlist, m, ksub, dsub, metric;
  uint32_t nbits, mb; /* bits per code (8 | 4), code bytes per row (m * nbits / 8) */
  uint64_t n_rows;
  float *centroids;   /* [nlist, dim] */
  float *cnorm;       /* [nlist] chain_dot(c,c) */
  float *codebook;    /* [m, ksub, dsub] */
  uint64_t *part_off; /* [nlist+1] */
  uint8_t *codes_t;   /* partition p: [mb, len_p] at mb*part_off[p] */
  uint64_t *row_ids;  /* [n_rows] */
  void *raw;          /* [n_rows, dim] or NULL */
  uint32_t raw_dtype;
  int borrowed;       /* codes_t / row_ids / raw point into caller memory */
} orc_index;

/* ------------------------------------------------------------------ chains */
float orc_chain_dot(const float *a, const float *b, uint32_t n) {
  float acc = 0.0f;
  for (uint32_t d = 0; d < n; ++d) acc = fmaf(a[d], b[d], acc);
  return acc;
}

float orc_chain_l2(const float *a, const float *b, uint32_t n) {
  float acc = 0.0f;
  for (uint32_t d = 0; d < n; ++d) {
    float t = a[d] - b[d];
    acc = fmaf(t, t, acc);
  }
  return acc;
}

static inline float bf16_to_f32(uint16_t h) {
  uint32_t u = ((uint32_t)h) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

static inline float f16_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1f;
  uint32_t man = h & 0x3ffu;
  uint32_t u;
  if (exp == 0) {
    if (man == 0) {
      u = sign;
    } else { /* subnormal */
      int e = -1;
      do {
        man <<= 1;
        ++e;
      } while (!(man & 0x400u));
      man &= 0x3ffu;
      u = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    u = sign | 0x7f800000u | (man << 13);
  } else {
    u = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* load row `i` of a raw column as f32 (exact widening) */
static void load_row(const void *base, uint32_t dtype, uint64_t i, uint32_t dim,
                     float *out) {
  if (dtype == MI355_DTYPE_F32) {
    memcpy(out, (const float *)base + i * dim, sizeof(float) * dim);
  } else if (dtype == MI355_DTYPE_BF16) {
    const uint16_t *p = (const uint16_t *)base + i * dim;
    for (uint32_t d = 0; d < dim; ++d) out[d] = bf16_to_f32(p[d]);
  } else {
    const uint16_t *p = (const uint16_t *)base + i * dim;
    for (uint32_t d = 0; d < dim; ++d) out[d] = f16_to_f32(p[d]);
  }
}

/* exact distance of the flat / refine path (lib.rs:236-260 metric meanings) */
float orc_exact_distance(const float *q, const float *v, uint32_t dim,
                         uint32_t metric) {
  if (metric == MI355_METRIC_L2) return orc_chain_l2(q, v, dim);
  if (metric == MI355_METRIC_DOT) return 1.0f - orc_chain_dot(q, v, dim);
  float qq = orc_chain_dot(q, q, dim);
  float vv = orc_chain_dot(v, v, dim);
  float qv = orc_chain_dot(q, v, dim);
  return 1.0f - qv / (sqrtf(qq) * sqrtf(vv));
}

/* ------------------------------------------------------------ top-k (E8) */
typedef struct {
  float d;
  uint64_t id;
  uint64_t pos; /* index position (for refine) */
} cand_t;

/* (distance ASC, rowid ASC): python/python/lancedb/query.py:1368 */
static inline int cand_less(const cand_t *a, const cand_t *b) {
  if (a->d < b->d) return 1;
  if (a->d > b->d) return 0;
  return a->id < b->id;
}

typedef struct {
  cand_t *h; /* max-heap on cand_less: h[0] is the worst kept */
  uint32_t n, cap;
} heap_t;

static void heap_sift_down(heap_t *hp, uint32_t i) {
  for (;;) {
    uint32_t l = 2 * i + 1, r = l + 1, w = i;
    if (l < hp->n && cand_less(&hp->h[w], &hp->h[l])) w = l;
    if (r < hp->n && cand_less(&hp->h[w], &hp->h[r])) w = r;
    if (w == i) return;
    cand_t t = hp->h[i];
    hp->h[i] = hp->h[w];
    hp->h[w] = t;
    i = w;
  }
}

static inline void heap_push(heap_t *hp, cand_t c) {
  if (hp->cap == 0) return;
  if (hp->n < hp->cap) {
    uint32_t i = hp->n++;
    hp->h[i] = c;
    while (i > 0) {
      uint32_t p = (i - 1) / 2;
      if (!cand_less(&hp->h[p], &hp->h[i])) break;
      cand_t t = hp->h[i];
      hp->h[i] = hp->h[p];
      hp->h[p] = t;
      i = p;
    }
  } else if (cand_less(&c, &hp->h[0])) {
    hp->h[0] = c;
    heap_sift_down(hp, 0);
  }
}

static int cand_cmp_qsort(const void *a, const void *b) {
  const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
  if (cand_less(x, y)) return -1;
  if (cand_less(y, x)) return 1;
  return 0;
}

/* distance_range [lower, upper) (query.rs:1282-1288); NaN = NULL, dropped
   (query.py:1370 FilterExec _distance IS NOT NULL) */
static inline int in_range(float d, const mi355_search_params *p) {
  if (d != d) return 0;
  if (p->has_lower_bound && !(d >= p->lower_bound)) return 0;
  if (p->has_upper_bound && !(d < p->upper_bound)) return 0;
  return 1;
}

/* prefilter (query.rs:489-507: prefilter = true is the default): the row must be
   on the allow list / off the block list; sorted u64 array, binary search */
static inline int row_permitted(uint64_t id, const mi355_search_params *p) {
  if (p->filter_mode == MI355_FILTER_NONE) return 1;
  uint64_t lo = 0, hi = p->n_filter;
  while (lo < hi) {
    uint64_t mid = lo + (hi - lo) / 2;
    if (p->filter_rowids[mid] < id) lo = mid + 1; else hi = mid;
  }
  int found = lo < p->n_filter && p->filter_rowids[lo] == id;
  return p->filter_mode == MI355_FILTER_ALLOW ? found : !found;
}

/* -------------------------------------------------------------- lifecycle */
int32_t orc_index_close(orc_index *ix) {
  if (!ix) return MI355_OK;
  free(ix->centroids);
  free(ix->cnorm);
  free(ix->codebook);
  free(ix->part_off);
  if (!ix->borrowed) {
    free(ix->codes_t);
    free(ix->row_ids);
    free(ix->raw);
  }
  free(ix);
  return MI355_OK;
}

/* borrow != 0: keep pointers to the caller's transposed codes / row ids / raw
   vectors instead of copying them (bench.py: a 9.6 GB code block). */
int32_t orc_index_open2(const mi355_index_desc *d, orc_index **out, int borrow);

int32_t orc_index_open(const mi355_index_desc *d, orc_index **out) {
  return orc_index_open2(d, out, 0);
}

int32_t orc_index_open2(const mi355_index_desc *d, orc_index **out, int borrow) {
  if (!d || !out || d->struct_size != sizeof(mi355_index_desc))
    return MI355_ERR_INVALID_INPUT;
  if (d->mem != MI355_MEM_HOST) return MI355_ERR_INVALID_INPUT;
  if (d->nbits != 8 && d->nbits != 4) return MI355_ERR_INVALID_INPUT;
  if (d->dim == 0 || d->m == 0 || d->dim % d->m != 0 || d->nlist == 0)
    return MI355_ERR_INVALID_INPUT;
  if (d->nbits == 4 && d->m % 2) return MI355_ERR_INVALID_INPUT; /* table/create_index.rs:96-101 */
  if (d->metric > MI355_METRIC_DOT) return MI355_ERR_INVALID_INPUT;
  if (!d->centroids || !d->codebook || !d->part_offsets ||
      (d->n_rows && !d->codes))
    return MI355_ERR_INVALID_INPUT;
  if (d->part_offsets[0] != 0 || d->part_offsets[d->nlist] != d->n_rows)
    return MI355_ERR_INVALID_INPUT;
  for (uint32_t p = 0; p < d->nlist; ++p)
    if (d->part_offsets[p + 1] < d->part_offsets[p])
      return MI355_ERR_INVALID_INPUT;

  if (borrow && (d->codes_layout != MI355_CODES_PART_TRANSPOSED || !d->row_ids))
    return MI355_ERR_INVALID_INPUT;
  orc_index *ix = (orc_index *)calloc(1, sizeof(orc_index));
  ix->borrowed = borrow;
  ix->dim = d->dim;
  ix->nlist = d->nlist;
  ix->m = d->m;
  ix->nbits = d->nbits;
  ix->ksub = 1u << d->nbits;
  ix->mb = d->m * d->nbits / 8;
  ix->dsub = d->dim / d->m;
  ix->metric = d->metric;
  ix->n_rows = d->n_rows;
  size_t nc = (size_t)d->nlist * d->dim;
  ix->centroids = (float *)malloc(sizeof(float) * nc);
  memcpy(ix->centroids, d->centroids, sizeof(float) * nc);
  ix->cnorm = (float *)malloc(sizeof(float) * d->nlist);
  for (uint32_t p = 0; p < d->nlist; ++p)
    ix->cnorm[p] = orc_chain_dot(ix->centroids + (size_t)p * d->dim,
                                 ix->centroids + (size_t)p * d->dim, d->dim);
  size_t ncb = (size_t)d->m * ix->ksub * ix->dsub;
  ix->codebook = (float *)malloc(sizeof(float) * ncb);
  memcpy(ix->codebook, d->codebook, sizeof(float) * ncb);
  ix->part_off = (uint64_t *)malloc(sizeof(uint64_t) * (d->nlist + 1));
  memcpy(ix->part_off, d->part_offsets, sizeof(uint64_t) * (d->nlist + 1));
  if (borrow) {
    ix->codes_t = (uint8_t *)d->codes;
    ix->row_ids = (uint64_t *)d->row_ids;
    ix->raw = (void *)d->raw_vectors;
    ix->raw_dtype = d->raw_dtype;
    *out = ix;
    return MI355_OK;
  }
  const uint32_t mb = ix->mb;
  ix->codes_t = (uint8_t *)malloc((size_t)d->n_rows * mb + 1);
  if (d->codes_layout == MI355_CODES_PART_TRANSPOSED) {
    memcpy(ix->codes_t, d->codes, (size_t)d->n_rows * mb);
  } else {
    for (uint32_t p = 0; p < d->nlist; ++p) {
      uint64_t o = ix->part_off[p], len = ix->part_off[p + 1] - o;
      uint8_t *dst = ix->codes_t + (size_t)o * mb;
      for (uint64_t i = 0; i < len; ++i)
        for (uint32_t j = 0; j < mb; ++j)
          dst[(size_t)j * len + i] = d->codes[(size_t)(o + i) * mb + j];
    }
  }
  ix->row_ids = (uint64_t *)malloc(sizeof(uint64_t) * (d->n_rows + 1));
  for (uint64_t i = 0; i < d->n_rows; ++i)
    ix->row_ids[i] = d->row_ids ? d->row_ids[i] : i;
  ix->raw_dtype = d->raw_dtype;
  if (d->raw_vectors) {
    size_t es = d->raw_dtype == MI355_DTYPE_F32 ? 4 : 2;
    ix->raw = malloc((size_t)d->n_rows * d->dim * es + 1);
    memcpy(ix->raw, d->raw_vectors, (size_t)d->n_rows * d->dim * es);
  }
  *out = ix;
  return MI355_OK;
}

/* ---------------------------------------------------------- stage kernels */

/* E2: coarse quantiser distances of one (already normalised, if cosine) query */
void orc_coarse(const orc_index *ix, const float *q, float *out /*[nlist]*/) {
  float qq = orc_chain_dot(q, q, ix->dim);
  for (uint32_t p = 0; p < ix->nlist; ++p) {
    float qc = orc_chain_dot(q, ix->centroids + (size_t)p * ix->dim, ix->dim);
    if (ix->metric == MI355_METRIC_DOT)
      out[p] = 1.0f - qc;
    else
      out[p] = fmaf(-2.0f, qc, qq + ix->cnorm[p]);
  }
}

/* The same coarse distances, eight centroids at a time: lane l of a 256-bit register carries centroid p + l's own
   d-ascending fmaf chain (the rows of an 8 x 8 block are transposed in registers, then eight vfmadd231ps with q[d]
   broadcast), so every lane computes exactly orc_chain_dot's sequence of roundings — bit-identical to orc_coarse
   (tests/test_oracle_golden.py compares the two with ==).  This is the "optimised path" SURVEY.md section 8d asks the
   CPU baseline to time; orc_coarse stays the reference the tests read. */
void orc_coarse_fast(const orc_index *ix, const float *q, float *out /*[nlist]*/) {
#ifdef __AVX2__
  const uint32_t dim = ix->dim, d8 = dim & ~7u;
  float qq = orc_chain_dot(q, q, dim);
  uint32_t p = 0;
  for (; p + 8 <= ix->nlist; p += 8) {
    const float *c = ix->centroids + (size_t)p * dim;
    __m256 acc = _mm256_setzero_ps();
    for (uint32_t d = 0; d < d8; d += 8) {
      __m256 r0 = _mm256_loadu_ps(c + 0 * (size_t)dim + d), r1 = _mm256_loadu_ps(c + 1 * (size_t)dim + d);
      __m256 r2 = _mm256_loadu_ps(c + 2 * (size_t)dim + d), r3 = _mm256_loadu_ps(c + 3 * (size_t)dim + d);
      __m256 r4 = _mm256_loadu_ps(c + 4 * (size_t)dim + d), r5 = _mm256_loadu_ps(c + 5 * (size_t)dim + d);
      __m256 r6 = _mm256_loadu_ps(c + 6 * (size_t)dim + d), r7 = _mm256_loadu_ps(c + 7 * (size_t)dim + d);
      /* 8 x 8 transpose: t[k] lane l = centroid p + l, element d + k */
      __m256 a0 = _mm256_unpacklo_ps(r0, r1), a1 = _mm256_unpackhi_ps(r0, r1);
      __m256 a2 = _mm256_unpacklo_ps(r2, r3), a3 = _mm256_unpackhi_ps(r2, r3);
      __m256 a4 = _mm256_unpacklo_ps(r4, r5), a5 = _mm256_unpackhi_ps(r4, r5);
      __m256 a6 = _mm256_unpacklo_ps(r6, r7), a7 = _mm256_unpackhi_ps(r6, r7);
      __m256 b0 = _mm256_shuffle_ps(a0, a2, 0x44), b1 = _mm256_shuffle_ps(a0, a2, 0xEE);
      __m256 b2 = _mm256_shuffle_ps(a1, a3, 0x44), b3 = _mm256_shuffle_ps(a1, a3, 0xEE);
      __m256 b4 = _mm256_shuffle_ps(a4, a6, 0x44), b5 = _mm256_shuffle_ps(a4, a6, 0xEE);
      __m256 b6 = _mm256_shuffle_ps(a5, a7, 0x44), b7 = _mm256_shuffle_ps(a5, a7, 0xEE);
      __m256 t0 = _mm256_permute2f128_ps(b0, b4, 0x20), t1 = _mm256_permute2f128_ps(b1, b5, 0x20);
      __m256 t2 = _mm256_permute2f128_ps(b2, b6, 0x20), t3 = _mm256_permute2f128_ps(b3, b7, 0x20);
      __m256 t4 = _mm256_permute2f128_ps(b0, b4, 0x31), t5 = _mm256_permute2f128_ps(b1, b5, 0x31);
      __m256 t6 = _mm256_permute2f128_ps(b2, b6, 0x31), t7 = _mm256_permute2f128_ps(b3, b7, 0x31);
      acc = _mm256_fmadd_ps(_mm256_set1_ps(q[d + 0]), t0, acc);
      acc = _mm256_fmadd_ps(_mm256_set1_ps(q[d + 1]), t1, acc);
      acc = _mm256_fmadd_ps(_mm256_set1_ps(q[d + 2]), t2, acc);
      acc = _mm256_fmadd_ps(_mm256_set1_ps(q[d + 3]), t3, acc);
      acc = _mm256_fmadd_ps(_mm256_set1_ps(q[d + 4]), t4, acc);
      acc = _mm256_fmadd_ps(_mm256_set1_ps(q[d + 5]), t5, acc);
      acc = _mm256_fmadd_ps(_mm256_set1_ps(q[d + 6]), t6, acc);
      acc = _mm256_fmadd_ps(_mm256_set1_ps(q[d + 7]), t7, acc);
    }
    float qc[8];
    _mm256_storeu_ps(qc, acc);
    for (uint32_t l = 0; l < 8; ++l) {
      float a = qc[l];
      for (uint32_t d = d8; d < dim; ++d) a = fmaf(q[d], c[(size_t)l * dim + d], a);  /* the chain's last dim % 8 steps */
      out[p + l] = ix->metric == MI355_METRIC_DOT ? 1.0f - a : fmaf(-2.0f, a, qq + ix->cnorm[p + l]);
    }
  }
  for (; p < ix->nlist; ++p) {
    float a = orc_chain_dot(q, ix->centroids + (size_t)p * dim, dim);
    out[p] = ix->metric == MI355_METRIC_DOT ? 1.0f - a : fmaf(-2.0f, a, qq + ix->cnorm[p]);
  }
#else
  orc_coarse(ix, q, out);
#endif
}

/* the nprobe smallest partitions by (distance, id); out sorted that way.
   NaN coarse distances sort last. */
void orc_select_probes(const float *dist, uint32_t nlist, uint32_t nprobe,
                       uint32_t *out) {
  cand_t *c = (cand_t *)malloc(sizeof(cand_t) * nlist);
  for (uint32_t p = 0; p < nlist; ++p) {
    c[p].d = dist[p] != dist[p] ? INFINITY : dist[p];
    c[p].id = dist[p] != dist[p] ? (uint64_t)nlist + p : p;
    c[p].pos = p;
  }
  qsort(c, nlist, sizeof(cand_t), cand_cmp_qsort);
  for (uint32_t i = 0; i < nprobe && i < nlist; ++i) out[i] = (uint32_t)c[i].pos;
  free(c);
}

/* E5: distance table of one (query, partition).  q is the preprocessed query. */
void orc_build_lut(const orc_index *ix, const float *q, uint32_t part,
                   float *lut /*[m,ksub]*/) {
  uint32_t dsub = ix->dsub, ks = ix->ksub;
  float r[dsub];
  for (uint32_t j = 0; j < ix->m; ++j) {
    const float *cb = ix->codebook + (size_t)j * ks * dsub;
    if (ix->metric == MI355_METRIC_DOT) {
      for (uint32_t c = 0; c < ks; ++c)
        lut[j * ks + c] = 1.0f - orc_chain_dot(q + j * dsub, cb + c * dsub, dsub);
    } else {
      const float *cen = ix->centroids + (size_t)part * ix->dim + j * dsub;
      for (uint32_t t = 0; t < dsub; ++t) r[t] = q[j * dsub + t] - cen[t];
      for (uint32_t c = 0; c < ks; ++c)
        lut[j * ks + c] = orc_chain_l2(r, cb + c * dsub, dsub);
    }
  }
}

static inline float finalize(const orc_index *ix, float acc) {
  if (ix->metric == MI355_METRIC_COSINE) return acc * 0.5f;
  if (ix->metric == MI355_METRIC_DOT) return acc - (float)(ix->m - 1);
  return acc;
}

/* E6: ADC over one partition; out[i] is the final approximate distance of the
   partition's i-th row.  Sub-quantiser-major sweep, plain f32 adds, j ascending. */
void orc_adc_partition(const orc_index *ix, const float *lut, uint32_t part,
                       float *out) {
  uint64_t o = ix->part_off[part], len = ix->part_off[part + 1] - o;
  const uint8_t *codes = ix->codes_t + (size_t)o * ix->mb;
  for (uint64_t i = 0; i < len; ++i) out[i] = 0.0f;
  if (ix->nbits == 8) {
    /* eight sub-quantisers per sweep of out[]: each row still adds its table values one by one in j order (the same
       roundings as one sweep per j), with an eighth of the out[] traffic */
    uint32_t j = 0;
    for (; j + 8 <= ix->m; j += 8) {
      const float *t = lut + (size_t)j * 256;
      const uint8_t *r0 = codes + (size_t)j * len;
      for (uint64_t i = 0; i < len; ++i) {
        float acc = out[i];
        acc = acc + t[0 * 256 + r0[0 * len + i]];
        acc = acc + t[1 * 256 + r0[1 * len + i]];
        acc = acc + t[2 * 256 + r0[2 * len + i]];
        acc = acc + t[3 * 256 + r0[3 * len + i]];
        acc = acc + t[4 * 256 + r0[4 * len + i]];
        acc = acc + t[5 * 256 + r0[5 * len + i]];
        acc = acc + t[6 * 256 + r0[6 * len + i]];
        acc = acc + t[7 * 256 + r0[7 * len + i]];
        out[i] = acc;
      }
    }
    for (; j < ix->m; ++j) {
      const float *t = lut + j * 256;
      const uint8_t *row = codes + (size_t)j * len;
      for (uint64_t i = 0; i < len; ++i) out[i] = out[i] + t[row[i]];
    }
  } else { /* byte t: sub-quantiser 2t in the low nibble, 2t+1 in the high nibble; adds stay j-ascending per row */
    for (uint32_t jb = 0; jb < ix->mb; ++jb) {
      const float *t0 = lut + (2 * jb) * 16, *t1 = t0 + 16;
      const uint8_t *row = codes + (size_t)jb * len;
      for (uint64_t i = 0; i < len; ++i) out[i] = out[i] + t0[row[i] & 15u];
      for (uint64_t i = 0; i < len; ++i) out[i] = out[i] + t1[row[i] >> 4];
    }
  }
  for (uint64_t i = 0; i < len; ++i) out[i] = finalize(ix, out[i]);
}

static void preprocess_query(const orc_index *ix, const float *q, float *qp) {
  if (ix->metric == MI355_METRIC_COSINE) {
    float nrm = sqrtf(orc_chain_dot(q, q, ix->dim));
    for (uint32_t d = 0; d < ix->dim; ++d) qp[d] = q[d] / nrm;
  } else {
    memcpy(qp, q, sizeof(float) * ix->dim);
  }
}

/* CPU seconds per stage, summed over the threads of the last orc_search call: coarse, probe select, LUT build, ADC,
   top-k heap, refine (orc_last_stage_seconds).  What bench.py prints beside the CPU baseline. */
#define ORC_N_STAGES 6
static double g_stage_seconds[ORC_N_STAGES];
static inline double orc_now(void) {
#ifdef _OPENMP
  return omp_get_wtime();
#else
  return 0.0;
#endif
}
void orc_last_stage_seconds(double *out /*[6]*/) { memcpy(out, g_stage_seconds, sizeof g_stage_seconds); }

/* one query over `nprobe` partitions -> heap of up to kk approximate hits */
static void search_one(const orc_index *ix, const float *q,
                       const mi355_search_params *prm, uint32_t nprobe,
                       uint32_t kk, heap_t *hp, float *scratch_lut,
                       float *scratch_dist, uint32_t *probes, float *coarse,
                       float *qp, uint64_t *n_scanned, double *stage /*[ORC_N_STAGES]*/) {
  double t0 = orc_now(), t1;
  preprocess_query(ix, q, qp);
  orc_coarse_fast(ix, qp, coarse);
  t1 = orc_now(); stage[0] += t1 - t0; t0 = t1;
  orc_select_probes(coarse, ix->nlist, nprobe, probes);
  t1 = orc_now(); stage[1] += t1 - t0; t0 = t1;
  hp->n = 0;
  hp->cap = kk;
  for (uint32_t pi = 0; pi < nprobe; ++pi) {
    uint32_t p = probes[pi];
    uint64_t o = ix->part_off[p], len = ix->part_off[p + 1] - o;
    if (len == 0) continue;
    t0 = orc_now();
    orc_build_lut(ix, qp, p, scratch_lut);
    t1 = orc_now(); stage[2] += t1 - t0; t0 = t1;
    orc_adc_partition(ix, scratch_lut, p, scratch_dist);
    t1 = orc_now(); stage[3] += t1 - t0; t0 = t1;
    if (n_scanned) *n_scanned += len;
    for (uint64_t i = 0; i < len; ++i) {
      float d = scratch_dist[i];
      if (!in_range(d, prm)) continue;
      if (!row_permitted(ix->row_ids[o + i], prm)) continue;
      cand_t c = {d, ix->row_ids[o + i], o + i};
      heap_push(hp, c);
    }
    stage[4] += orc_now() - t0;
  }
}

static uint64_t max_part_len(const orc_index *ix) {
  uint64_t mx = 1;
  for (uint32_t p = 0; p < ix->nlist; ++p) {
    uint64_t l = ix->part_off[p + 1] - ix->part_off[p];
    if (l > mx) mx = l;
  }
  return mx;
}

static int32_t check_params(const mi355_search_params *prm, uint32_t index_metric) {
  if (!prm || prm->struct_size != sizeof(mi355_search_params))
    return MI355_ERR_INVALID_INPUT;
  if (prm->nprobe_min == 0) return MI355_ERR_INVALID_INPUT; /* query.rs:1232-1236 */
  if (prm->nprobe_max != 0 && prm->nprobe_max < prm->nprobe_min)
    return MI355_ERR_INVALID_INPUT; /* query.rs:1237-1244, :1269-1275 */
  if (prm->metric != MI355_METRIC_DEFAULT && prm->metric != index_metric)
    return MI355_ERR_INVALID_INPUT; /* query.rs:1347-1349: must match the index */
  return MI355_OK;
}

/*
 * IVF-PQ search of n_queries queries (E2+E3+E8+E9).  nthreads: one query per
 * thread (the reference searches on the tokio worker threads,
 * python/src/runtime.rs:31-37); 0 = all cores.
 */
int32_t orc_search(const orc_index *ix, const float *queries, uint32_t n_queries,
                   const mi355_search_params *prm, uint64_t *out_rowids,
                   float *out_dist, uint32_t *out_counts, int32_t nthreads,
                   uint64_t *out_vectors_scanned) {
  if (!ix) return MI355_ERR_INVALID_INPUT;
  int32_t st = check_params(prm, ix->metric);
  if (st) return st;
  uint32_t k = prm->k;
  if (n_queries && (!queries || !out_counts || (k && (!out_rowids || !out_dist))))
    return MI355_ERR_INVALID_INPUT;
  if (prm->refine_factor && !ix->raw) return MI355_ERR_INVALID_INPUT;
  uint32_t np_min = prm->nprobe_min > ix->nlist ? ix->nlist : prm->nprobe_min;
  uint32_t np_max = (prm->nprobe_max == 0 || prm->nprobe_max > ix->nlist)
                        ? ix->nlist
                        : prm->nprobe_max;
  uint64_t kk64 = (uint64_t)k * (prm->refine_factor ? prm->refine_factor : 1);
  uint32_t kk = kk64 > ix->n_rows ? (uint32_t)ix->n_rows : (uint32_t)kk64;
  uint64_t mpl = max_part_len(ix);
  uint64_t total_scanned = 0;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
  nthreads = 1;
#endif
  memset(g_stage_seconds, 0, sizeof g_stage_seconds);
#pragma omp parallel num_threads(nthreads) reduction(+ : total_scanned)
  {
    double stage[ORC_N_STAGES] = {0, 0, 0, 0, 0, 0};
    float *lut = (float *)malloc(sizeof(float) * ix->m * ix->ksub);
    float *dist = (float *)malloc(sizeof(float) * mpl);
    float *coarse = (float *)malloc(sizeof(float) * ix->nlist);
    uint32_t *probes = (uint32_t *)malloc(sizeof(uint32_t) * ix->nlist);
    float *qp = (float *)malloc(sizeof(float) * ix->dim);
    float *rawrow = (float *)malloc(sizeof(float) * ix->dim);
    heap_t hp;
    hp.h = (cand_t *)malloc(sizeof(cand_t) * (kk + 1));
#pragma omp for schedule(dynamic, 1)
    for (uint32_t qi = 0; qi < n_queries; ++qi) {
      const float *q = queries + (size_t)qi * ix->dim;
      uint64_t scanned = 0;
      search_one(ix, q, prm, np_min, kk, &hp, lut, dist, probes, coarse, qp,
                 &scanned, stage);
      /* maximum_nprobes (query.rs:1246-1262): the excess partitions are
         searched only if the first pass found fewer than k rows */
      if (hp.n < kk && np_max > np_min) {
        scanned = 0;
        search_one(ix, q, prm, np_max, kk, &hp, lut, dist, probes, coarse, qp,
                   &scanned, stage);
      }
      total_scanned += scanned;
      if (prm->refine_factor) {
        /* E9: exact distance on the raw vectors of the kk candidates, range
           filter on the exact distance, re-sort */
        double tr = orc_now();
        uint32_t w = 0;
        for (uint32_t i = 0; i < hp.n; ++i) {
          load_row(ix->raw, ix->raw_dtype, hp.h[i].pos, ix->dim, rawrow);
          float d = orc_exact_distance(q, rawrow, ix->dim, ix->metric);
          if (!in_range(d, prm)) continue;
          hp.h[w] = hp.h[i];
          hp.h[w].d = d;
          ++w;
        }
        hp.n = w;
        stage[5] += orc_now() - tr;
      }
      qsort(hp.h, hp.n, sizeof(cand_t), cand_cmp_qsort);
      uint32_t cnt = hp.n < k ? hp.n : k;
      out_counts[qi] = cnt;
      for (uint32_t i = 0; i < k; ++i) {
        out_rowids[(size_t)qi * k + i] = i < cnt ? hp.h[i].id : UINT64_MAX;
        out_dist[(size_t)qi * k + i] = i < cnt ? hp.h[i].d : INFINITY;
      }
    }
    free(lut);
    free(dist);
    free(coarse);
    free(probes);
    free(qp);
    free(rawrow);
    free(hp.h);
#pragma omp critical
    for (int s2 = 0; s2 < ORC_N_STAGES; ++s2) g_stage_seconds[s2] += stage[s2];
  }
  if (out_vectors_scanned) *out_vectors_scanned = total_scanned;
  return MI355_OK;
}

/*
 * E7 + E8: flat KNN over a raw column (KNNVectorDistance + TopK,
 * python/python/lancedb/query.py:1365-1370).
 */
int32_t orc_flat_search(const mi355_flat_desc *fd, const float *queries,
                        uint32_t n_queries, const mi355_search_params *prm,
                        uint64_t *out_rowids, float *out_dist,
                        uint32_t *out_counts, int32_t nthreads) {
  if (!fd || fd->struct_size != sizeof(mi355_flat_desc) || fd->dim == 0)
    return MI355_ERR_INVALID_INPUT;
  if (fd->mem != MI355_MEM_HOST || fd->dtype > MI355_DTYPE_F16)
    return MI355_ERR_INVALID_INPUT;
  if (!prm || prm->struct_size != sizeof(mi355_search_params))
    return MI355_ERR_INVALID_INPUT;
  uint32_t metric = prm->metric == MI355_METRIC_DEFAULT ? MI355_METRIC_L2 : prm->metric;
  if (metric > MI355_METRIC_DOT) return MI355_ERR_INVALID_INPUT;
  uint32_t k = prm->k;
  uint32_t kk = k > fd->n_rows ? (uint32_t)fd->n_rows : k;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
  nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
  {
    float *row = (float *)malloc(sizeof(float) * fd->dim);
    heap_t hp;
    hp.h = (cand_t *)malloc(sizeof(cand_t) * (kk + 1));
#pragma omp for schedule(dynamic, 1)
    for (uint32_t qi = 0; qi < n_queries; ++qi) {
      const float *q = queries + (size_t)qi * fd->dim;
      hp.n = 0;
      hp.cap = kk;
      for (uint64_t i = 0; i < fd->n_rows; ++i) {
        load_row(fd->vectors, fd->dtype, i, fd->dim, row);
        float d = orc_exact_distance(q, row, fd->dim, metric);
        if (!in_range(d, prm)) continue;
        if (!row_permitted(fd->row_ids ? fd->row_ids[i] : i, prm)) continue;
        cand_t c = {d, fd->row_ids ? fd->row_ids[i] : i, i};
        heap_push(&hp, c);
      }
      qsort(hp.h, hp.n, sizeof(cand_t), cand_cmp_qsort);
      uint32_t cnt = hp.n < k ? hp.n : k;
      out_counts[qi] = cnt;
      for (uint32_t i = 0; i < k; ++i) {
        out_rowids[(size_t)qi * k + i] = i < cnt ? hp.h[i].id : UINT64_MAX;
        out_dist[(size_t)qi * k + i] = i < cnt ? hp.h[i].d : INFINITY;
      }
    }
    free(row);
    free(hp.h);
  }
  return MI355_OK;
}

/* k-way merge used to check mi355_merge_topk: lists [n_lists, n_queries, k] */
int32_t orc_merge_topk(const uint64_t *in_rowids, const float *in_dist,
                       const uint32_t *in_counts, uint32_t n_lists,
                       uint32_t n_queries, uint32_t k, uint64_t *out_rowids,
                       float *out_dist, uint32_t *out_counts) {
  cand_t *c = (cand_t *)malloc(sizeof(cand_t) * ((size_t)n_lists * k + 1));
  for (uint32_t q = 0; q < n_queries; ++q) {
    uint32_t n = 0;
    for (uint32_t l = 0; l < n_lists; ++l) {
      uint32_t cnt = in_counts[(size_t)l * n_queries + q];
      if (cnt > k) cnt = k;
      for (uint32_t i = 0; i < cnt; ++i) {
        size_t o = ((size_t)l * n_queries + q) * k + i;
        c[n].d = in_dist[o];
        c[n].id = in_rowids[o];
        c[n].pos = 0;
        ++n;
      }
    }
    qsort(c, n, sizeof(cand_t), cand_cmp_qsort);
    uint32_t cnt = n < k ? n : k;
    out_counts[q] = cnt;
    for (uint32_t i = 0; i < k; ++i) {
      out_rowids[(size_t)q * k + i] = i < cnt ? c[i].id : UINT64_MAX;
      out_dist[(size_t)q * k + i] = i < cnt ? c[i].d : INFINITY;
    }
  }
  free(c);
  return MI355_OK;
}

static int shard_cmp_qsort(const void *a, const void *b) {
  const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
  if (x->pos != y->pos) return x->pos > y->pos ? -1 : 1;
  if (x->id != y->id) return x->id < y->id ? -1 : 1;
  return 0;
}

/* same greedy plan as mi355_shard_plan (include/mi355_ann.h) */
int32_t orc_shard_plan(const uint64_t *part_offsets, uint32_t nlist,
                       uint32_t shard_count, uint32_t *out_owner) {
  if (!part_offsets || !out_owner || shard_count == 0) return MI355_ERR_INVALID_INPUT;
  cand_t *c = (cand_t *)malloc(sizeof(cand_t) * (nlist + 1));
  for (uint32_t p = 0; p < nlist; ++p) {
    uint64_t len = part_offsets[p + 1] - part_offsets[p];
    c[p].d = 0.0f; /* unused */
    c[p].id = p;
    c[p].pos = len;
  }
  qsort(c, nlist, sizeof(cand_t), shard_cmp_qsort); /* length DESC, id ASC */
  uint64_t *load = (uint64_t *)calloc(shard_count, sizeof(uint64_t));
  for (uint32_t i = 0; i < nlist; ++i) {
    uint32_t best = 0;
    for (uint32_t s = 1; s < shard_count; ++s)
      if (load[s] < load[best]) best = s;
    out_owner[c[i].id] = best;
    load[best] += c[i].pos;
  }
  free(load);
  free(c);
  return MI355_OK;
}

/* ------------------------------------------------------------ index population
 * Restates the transform stage of the index build (assign + residual PQ encode +
 * stable partition order); definitions in include/mi355_ann.h (mi355_ivfpq_encode). */
int32_t orc_ivfpq_encode(const mi355_encode_desc *d, const float *vectors, uint64_t n_rows,
                         uint64_t *out_part_offsets, uint8_t *out_codes, uint64_t *out_order,
                         uint32_t *out_assign) {
  if (!d || d->struct_size != sizeof(mi355_encode_desc) || !vectors || !out_part_offsets || !out_codes ||
      !out_order || (d->nbits != 8 && d->nbits != 4) || d->dim == 0 || d->m == 0 || d->dim % d->m ||
      d->metric > MI355_METRIC_DOT || (d->nbits == 4 && d->m % 2))
    return MI355_ERR_INVALID_INPUT;
  const uint32_t dim = d->dim, nlist = d->nlist, m = d->m, dsub = dim / m;
  const uint32_t ks = 1u << d->nbits, mb = m * d->nbits / 8; /* codebook entries, code bytes per row */
  float *cn = (float *)malloc(sizeof(float) * nlist);
  for (uint32_t p = 0; p < nlist; ++p)
    cn[p] = orc_chain_dot(d->centroids + (size_t)p * dim, d->centroids + (size_t)p * dim, dim);
  uint32_t *assign = (uint32_t *)malloc(sizeof(uint32_t) * (n_rows ? n_rows : 1));
  uint8_t *codes_src = (uint8_t *)malloc((size_t)m * (n_rows ? n_rows : 1));
#pragma omp parallel
  {
    float *x = (float *)malloc(sizeof(float) * dim);
#pragma omp for schedule(static)
    for (uint64_t i = 0; i < n_rows; ++i) {
      const float *src = vectors + (size_t)i * dim;
      float qq = orc_chain_dot(src, src, dim);
      if (d->metric == MI355_METRIC_COSINE) {
        float nrm = sqrtf(qq);
        for (uint32_t t = 0; t < dim; ++t) x[t] = src[t] / nrm;
        qq = orc_chain_dot(x, x, dim);
      } else {
        memcpy(x, src, sizeof(float) * dim);
      }
      uint32_t best = 0;
      float bd = 0.0f;
      int have = 0;
      for (uint32_t p = 0; p < nlist; ++p) {
        float dot = orc_chain_dot(x, d->centroids + (size_t)p * dim, dim);
        float dd = d->metric == MI355_METRIC_DOT ? 1.0f - dot : fmaf(-2.0f, dot, qq + cn[p]);
        /* NaN never wins; the first finite-or-inf minimum does */
        if (dd == dd && (!have || dd < bd)) {
          bd = dd;
          best = p;
          have = 1;
        }
      }
      assign[i] = best;
      if (d->metric != MI355_METRIC_DOT)
        for (uint32_t t = 0; t < dim; ++t) x[t] = x[t] - d->centroids[(size_t)best * dim + t];
      for (uint32_t j = 0; j < m; ++j) {
        uint32_t bc = 0;
        float bv = 0.0f;
        int hv = 0;
        for (uint32_t c = 0; c < ks; ++c) {
          const float *cb = d->codebook + ((size_t)j * ks + c) * dsub;
          float v = d->metric == MI355_METRIC_DOT ? 1.0f - orc_chain_dot(x + j * dsub, cb, dsub)
                                                  : orc_chain_l2(x + j * dsub, cb, dsub);
          if (v == v && (!hv || v < bv)) {
            bv = v;
            bc = c;
            hv = 1;
          }
        }
        codes_src[(size_t)i * m + j] = (uint8_t)bc;
      }
    }
    free(x);
  }
  uint64_t *cnt = (uint64_t *)calloc(nlist + 1, sizeof(uint64_t));
  for (uint64_t i = 0; i < n_rows; ++i) cnt[assign[i] + 1]++;
  for (uint32_t p = 0; p < nlist; ++p) cnt[p + 1] += cnt[p];
  memcpy(out_part_offsets, cnt, sizeof(uint64_t) * (nlist + 1));
  for (uint64_t i = 0; i < n_rows; ++i) { /* stable: source order inside a partition */
    uint64_t pos = cnt[assign[i]]++;
    out_order[pos] = i;
    if (d->nbits == 8) {
      memcpy(out_codes + (size_t)pos * m, codes_src + (size_t)i * m, m);
    } else { /* pack two codes per byte: 2t low nibble, 2t+1 high nibble */
      for (uint32_t t = 0; t < mb; ++t)
        out_codes[(size_t)pos * mb + t] =
            (uint8_t)(codes_src[(size_t)i * m + 2 * t] | (codes_src[(size_t)i * m + 2 * t + 1] << 4));
    }
  }
  if (out_assign) memcpy(out_assign, assign, sizeof(uint32_t) * n_rows);
  free(cnt);
  free(codes_src);
  free(assign);
  free(cn);
  return MI355_OK;
}

/* ---- index training (checker of mi355_kmeans_train / mi355_ivf_residuals) ----------
 * Deterministic Lloyd iteration as defined in include/mi355_ann.h; the reference's
 * build parameters: rust/lancedb/src/index/vector.rs:61-119 (num_partitions,
 * max_iterations, sample_rate, num_sub_vectors), create_index.rs:283-303.  lance's
 * trainer itself is an external dependency with a random initialisation [EXT]; what is
 * restated here is the definition both sides of the parity test follow. */
static void orc_prep_row(const float *src, uint32_t dim, uint32_t metric, float *x, float *qq) {
  float s = orc_chain_dot(src, src, dim);
  if (metric == MI355_METRIC_COSINE) {
    float nrm = sqrtf(s);
    for (uint32_t t = 0; t < dim; ++t) x[t] = src[t] / nrm;
    s = orc_chain_dot(x, x, dim);
  } else {
    memcpy(x, src, sizeof(float) * dim);
  }
  *qq = s;
}

static uint32_t orc_nearest(const float *x, float qq, const float *cen, const float *cn, uint32_t k,
                            uint32_t dim, uint32_t metric) {
  uint32_t best = 0;
  float bd = 0.0f;
  int have = 0;
  for (uint32_t p = 0; p < k; ++p) {
    float dot = orc_chain_dot(x, cen + (size_t)p * dim, dim);
    float dd = metric == MI355_METRIC_DOT ? 1.0f - dot : fmaf(-2.0f, dot, qq + cn[p]);
    if (dd == dd && (!have || dd < bd)) {
      bd = dd;
      best = p;
      have = 1;
    }
  }
  return best;
}

int32_t orc_kmeans_train(const mi355_kmeans_desc *d, const float *vectors, uint64_t n_rows, float *centroids,
                         uint64_t *out_counts) {
  if (!d || d->struct_size != sizeof(mi355_kmeans_desc) || !centroids || d->dim == 0 || d->k == 0 ||
      d->metric > MI355_METRIC_DOT || (n_rows && !vectors))
    return MI355_ERR_INVALID_INPUT;
  const uint32_t dim = d->dim, k = d->k;
  const uint64_t ld = d->ld ? d->ld : dim;
  if (ld < dim) return MI355_ERR_INVALID_INPUT;
  float *x = (float *)malloc(sizeof(float) * dim * (n_rows ? n_rows : 1));
  float *qq = (float *)malloc(sizeof(float) * (n_rows ? n_rows : 1));
  uint32_t *assign = (uint32_t *)malloc(sizeof(uint32_t) * (n_rows ? n_rows : 1));
  float *cn = (float *)malloc(sizeof(float) * k);
  float *sum = (float *)malloc(sizeof(float) * (size_t)k * dim);
  uint64_t *cnt = (uint64_t *)calloc(k, sizeof(uint64_t));
#pragma omp parallel for schedule(static)
  for (uint64_t i = 0; i < n_rows; ++i) orc_prep_row(vectors + i * ld, dim, d->metric, x + i * dim, qq + i);
  for (uint32_t it = 0; it < d->iters; ++it) {
    for (uint32_t p = 0; p < k; ++p)
      cn[p] = orc_chain_dot(centroids + (size_t)p * dim, centroids + (size_t)p * dim, dim);
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < n_rows; ++i)
      assign[i] = orc_nearest(x + i * dim, qq[i], centroids, cn, k, dim, d->metric);
    memset(sum, 0, sizeof(float) * (size_t)k * dim);
    memset(cnt, 0, sizeof(uint64_t) * k);
    for (uint64_t i = 0; i < n_rows; ++i) { /* source-row order inside every centroid */
      float *s = sum + (size_t)assign[i] * dim;
      const float *r = x + i * dim;
      for (uint32_t t = 0; t < dim; ++t) s[t] = s[t] + r[t];
      cnt[assign[i]]++;
    }
    for (uint32_t p = 0; p < k; ++p)
      if (cnt[p]) {
        const float c = (float)cnt[p];
        for (uint32_t t = 0; t < dim; ++t) centroids[(size_t)p * dim + t] = sum[(size_t)p * dim + t] / c;
      }
  }
  if (out_counts) memcpy(out_counts, cnt, sizeof(uint64_t) * k);
  free(cnt);
  free(sum);
  free(cn);
  free(assign);
  free(qq);
  free(x);
  return MI355_OK;
}

int32_t orc_ivf_residuals(const mi355_kmeans_desc *d, const float *vectors, uint64_t n_rows, const float *centroids,
                          float *out_residuals, uint32_t *out_assign) {
  if (!d || d->struct_size != sizeof(mi355_kmeans_desc) || !centroids || d->dim == 0 || d->k == 0 ||
      d->metric > MI355_METRIC_DOT || (n_rows && (!vectors || !out_residuals)))
    return MI355_ERR_INVALID_INPUT;
  const uint32_t dim = d->dim, k = d->k;
  const uint64_t ld = d->ld ? d->ld : dim;
  float *cn = (float *)malloc(sizeof(float) * k);
  for (uint32_t p = 0; p < k; ++p)
    cn[p] = orc_chain_dot(centroids + (size_t)p * dim, centroids + (size_t)p * dim, dim);
#pragma omp parallel for schedule(static)
  for (uint64_t i = 0; i < n_rows; ++i) {
    float *x = out_residuals + i * dim, qq;
    orc_prep_row(vectors + i * ld, dim, d->metric, x, &qq);
    uint32_t a = orc_nearest(x, qq, centroids, cn, k, dim, d->metric);
    if (out_assign) out_assign[i] = a;
    if (d->metric != MI355_METRIC_DOT)
      for (uint32_t t = 0; t < dim; ++t) x[t] = x[t] - centroids[(size_t)a * dim + t];
  }
  free(cn);
  return MI355_OK;
}
