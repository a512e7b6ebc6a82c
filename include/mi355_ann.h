/*
 * mi355_ann.h — C ABI of the MI355X (gfx950) ANN scan engine.
 *
 * This is the drop-in boundary for LanceDB's vector-search hot path.  In the
 * reference there is no FFI on this path: `lancedb::query::VectorQuery` holds an
 * `Arc<dyn BaseTable>` (rust/lancedb/src/table.rs:549-576) and
 * `table::query::create_plan` (rust/lancedb/src/table/query.rs:131-328) hands a
 * `VectorQueryRequest` (rust/lancedb/src/query.rs:1066-1095) to
 * `lance::dataset::scanner::Scanner::{nearest, minimum_nprobes, maximum_nprobes,
 * refine, distance_metric, distance_range, use_index}` and then `create_plan()`
 * (table/query.rs:231-327).  The entry points below are what a Rust shim that
 * replaces the `ANNIvfPartitionExec` / `ANNIvfSubIndexExec` / `KNNVectorDistance`
 * plan nodes (named at table/query.rs:1079, python/python/lancedb/query.py:1369)
 * would bind with `extern "C"`.  See INTEGRATION.md for that shim.
 *
 * Conventions
 *   - every function returns an int32 status (0 = ok); no exception or abort
 *     crosses the ABI.  Codes map onto lancedb::Error variants
 *     (rust/lancedb/src/error.rs): 1 InvalidInput, 2 Runtime, 3 Timeout,
 *     4 NotSupported.
 *   - the caller owns every buffer it passes in or receives results in; the
 *     library copies index data to the device at open and never frees caller
 *     memory.  Handles are opaque and released only by *_close.
 *   - plain pointers and sizes only: no torch / Arrow types.
 *   - result contract (python/python/lancedb/query.py:1365-1370): per query up to
 *     k rows sorted by (_distance ASC, _rowid ASC), NaN distances dropped,
 *     distances restricted to [lower_bound, upper_bound)
 *     (rust/lancedb/src/query.rs:1282-1288).  `_distance` is f32, `_rowid` u64
 *     (rust/lancedb/src/query/hybrid.rs:95-100).
 */
#ifndef MI355_ANN_H
#define MI355_ANN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_ANN_ABI_VERSION 5u

/* ---- status codes (rust/lancedb/src/error.rs:55-145) -------------------- */
enum {
  MI355_OK = 0,
  MI355_ERR_INVALID_INPUT = 1, /* lancedb::Error::InvalidInput */
  MI355_ERR_RUNTIME = 2,       /* lancedb::Error::Runtime (HIP failure text) */
  MI355_ERR_TIMEOUT = 3,       /* lancedb::Error::Timeout */
  MI355_ERR_NOT_SUPPORTED = 4  /* lancedb::Error::NotSupported */
};

/* ---- DistanceType (rust/lancedb/src/lib.rs:233-260) ---------------------- */
enum {
  MI355_METRIC_L2 = 0,     /* squared euclidean, [0, inf) */
  MI355_METRIC_COSINE = 1, /* 1 - cos, [0, 2] */
  MI355_METRIC_DOT = 2,    /* 1 - dot (lance_linalg dot_distance) */
  MI355_METRIC_DEFAULT = 255 /* "not set": use the metric the index was trained with
                                (VectorQueryRequest.distance_type = None) */
};

/* where the buffers named by a descriptor / call live */
enum {
  MI355_MEM_HOST = 0,
  MI355_MEM_DEVICE = 1 /* pointers are device pointers on `device` */
};

/* element type of raw vector columns (utils/mod.rs:289-298: any float type) */
enum {
  MI355_DTYPE_F32 = 0,
  MI355_DTYPE_BF16 = 1,
  MI355_DTYPE_F16 = 2
};

/* ApproxMode (rust/lancedb/src/lib.rs:298-313; VectorQueryRequest.approx_mode,
   query.rs:1092, forwarded at table/query.rs:241-242).  "This currently only affects
   RQ-quantized vector indexes ... Other index types ignore this setting": carried,
   validated and ignored by the IVF-PQ / flat paths. */
enum {
  MI355_APPROX_UNSET = 0, /* approx_mode = None */
  MI355_APPROX_FAST = 1,
  MI355_APPROX_NORMAL = 2, /* the reference's default */
  MI355_APPROX_ACCURATE = 3
};

/* mi355_index_desc.flags */
enum {
  /* keep the generic [sub-quantiser][rows] code layout (k_scan_pair) even when the
     pre-skewed production layout is available for this m */
  MI355_INDEX_GENERIC_SCAN = 1u,
  /* raw_vectors stay in the CALLER's host memory (which must outlive the handle): the
     library page-locks the range and the refine kernel gathers the k*refine_factor
     candidate rows over PCIe (zero copy).  For columns that do not fit HBM (C5:
     100 M x 1536).  Requires mem == MI355_MEM_HOST. */
  MI355_INDEX_RAW_HOST_MAPPED = 2u,
  /* shard handles: codes, row_ids and raw_vectors hold ONLY the partitions this shard owns
     (mi355_shard_plan), concatenated in partition order — a rank never materialises the
     partitions of the others.  part_offsets stays the GLOBAL array (it defines the plan and
     the global positions); row_ids must then be given (identity ids would be global). */
  MI355_INDEX_LOCAL_ARRAYS = 4u
};

/* layout of the PQ code block handed to mi355_index_open */
enum {
  /* [n_rows, mb] u8, rows ordered by partition (partition p = rows
     part_offsets[p] .. part_offsets[p+1]); mb = m * nbits / 8 code bytes per row.
     4-bit codes: byte t of a row holds sub-quantiser 2t in its LOW nibble and 2t+1 in
     its HIGH nibble ([EXT] assumption, the common packing) */
  MI355_CODES_ROW_MAJOR = 0,
  /* per partition a [mb, len_p] u8 block (sub-quantiser major), partitions
     concatenated at byte offset mb * part_offsets[p]: the transposed storage
     lance-index keeps per partition ([EXT]) */
  MI355_CODES_PART_TRANSPOSED = 1
};

typedef struct mi355_index mi355_index; /* IVF-PQ index resident on one GPU */
typedef struct mi355_flat mi355_flat;   /* raw vector column resident on one GPU */

/*
 * Describes a trained IVF-PQ index.  Field meanings follow
 * IvfPqIndexBuilder -> IvfBuildParams / PQBuildParams
 * (rust/lancedb/src/index/vector.rs:266-319, table/create_index.rs:68-102,
 * :283-303).
 */
typedef struct mi355_index_desc {
  uint32_t struct_size; /* sizeof(mi355_index_desc), ABI guard */
  uint32_t dim;         /* vector dimension */
  uint32_t nlist;       /* IVF partitions (num_partitions) */
  uint32_t m;           /* PQ sub-vectors (num_sub_vectors); dim % m == 0 */
  uint32_t nbits;       /* PQ bits: 8, or 4 with even m (table/create_index.rs:96-101).
                           4-BIT PARITY AGAINST lance-index IS UNKNOWN: the engine sums f32 table entries per row like the
                           8-bit path (and matches this repository's CPU restatement bit for bit); lance-index's 4-bit scan
                           is, to our knowledge, a SIMD-shuffle scan over a table QUANTISED to u8 [EXT], whose distances
                           and tie-breaks differ materially.  No "bit-exact" statement of this library covers 4-bit
                           indexes against the reference; expect recall-level agreement, not id-level. */
  uint32_t metric;      /* MI355_METRIC_* the index was trained with */
  uint64_t n_rows;      /* rows covered by the index */
  uint32_t mem;         /* MI355_MEM_*: where the pointers below live */
  uint32_t codes_layout;/* MI355_CODES_* */
  const float *centroids;       /* [nlist, dim] f32 */
  const float *codebook;        /* [m, 2^nbits, dim/m] f32 */
  const uint64_t *part_offsets; /* [nlist+1] row offsets, ALWAYS host memory */
  const uint8_t *codes;         /* see codes_layout */
  const uint64_t *row_ids;      /* [n_rows] _rowid per indexed row, in index order;
                                   NULL = identity (row i has _rowid i) */
  const void *raw_vectors;      /* optional [n_rows, dim] raw vectors in index
                                   order (refine stage, query.rs:1302-1332);
                                   NULL = refine unavailable */
  uint32_t raw_dtype;           /* MI355_DTYPE_* of raw_vectors */
  int32_t device;               /* HIP device ordinal */
  /* Partition sharding over the GPUs of one node.  This
     handle keeps the partitions a greedy bytes-balanced bin-packing assigns to
     `shard_rank`; all other partitions are empty on this handle.  1/0 = no
     sharding. */
  uint32_t shard_count;
  uint32_t shard_rank;
  uint32_t flags;               /* MI355_INDEX_* */
  uint32_t reserved;
  /* optional [nlist] owner (shard id) of every partition, ALWAYS host memory: replaces the built-in plan,
     e.g. with mi355_shard_plan_weighted over observed probe counts.  Every rank must pass the same array.
     NULL = mi355_shard_plan(part_offsets, nlist, shard_count). */
  const uint32_t *part_owner;
} mi355_index_desc;

/*
 * Per-search parameters: the numeric subset of VectorQueryRequest +
 * QueryRequest (rust/lancedb/src/query.rs:1066-1114, :818-907).
 */
typedef struct mi355_search_params {
  uint32_t struct_size;   /* sizeof(mi355_search_params) */
  uint32_t k;             /* limit + offset (table/query.rs:231) */
  uint32_t nprobe_min;    /* minimum_nprobes (default 20) */
  uint32_t nprobe_max;    /* maximum_nprobes; 0 = None = all partitions */
  uint32_t refine_factor; /* 0 = not set; n>=1 fetches k*n candidates and
                             re-ranks by true distance (query.rs:1313-1317) */
  uint32_t metric;        /* MI355_METRIC_DEFAULT or must equal the index metric */
  uint32_t has_lower_bound;
  uint32_t has_upper_bound;
  float lower_bound;      /* inclusive */
  float upper_bound;      /* exclusive */
  uint32_t io_mem;        /* MI355_MEM_*: where queries / outputs live.  DEVICE
                             = enqueue on the handle's stream and return
                             without waiting (see mi355_*_set_stream): the results
                             are ordered on that stream — except in the two opt-in
                             overlap modes, whose results are complete at
                             mi355_index_sync (MI355_CFG_DEFER_REFINE for mi355_search,
                             the default overlap of mi355_search_sharded) */
  uint32_t timeout_ms;    /* 0 = none (QueryExecutionOptions.timeout, query.rs:641).  A device-side
                             deadline: the scan kernels stop taking work items once it has passed and
                             every later kernel of the call exits at once; host-I/O calls then return
                             MI355_ERR_TIMEOUT, device-I/O calls report it through mi355_last_stats
                             (timed_out) / mi355_index_sync.  Results of a timed-out call are undefined. */
  /* Prefilter (QueryRequest.filter with prefilter = true, the reference's default:
     rust/lancedb/src/query.rs:489-507, :899; table/query.rs:251-262): the rows a
     filter kept or dropped, as the caller's evaluation of the predicate, in the
     form lance passes to its ANN nodes (RowIdMask allow / block list [EXT]).
     `filter_rowids` is sorted ascending, lives where io_mem says, and is applied
     BEFORE the top-k: the result is the k nearest among the permitted rows. */
  uint32_t filter_mode;   /* MI355_FILTER_* */
  uint32_t approx_mode;   /* MI355_APPROX_* (ignored by IVF-PQ / flat, see the enum) */
  const uint64_t *filter_rowids; /* [n_filter] sorted ascending, unique */
  uint64_t n_filter;
} mi355_search_params;

enum {
  MI355_FILTER_NONE = 0,
  MI355_FILTER_ALLOW = 1, /* only rows whose _rowid is listed */
  MI355_FILTER_BLOCK = 2  /* every row except the listed ones */
};

typedef struct mi355_flat_desc {
  uint32_t struct_size;
  uint32_t dim;
  uint64_t n_rows;
  uint32_t dtype;          /* MI355_DTYPE_* */
  uint32_t mem;            /* MI355_MEM_* */
  const void *vectors;     /* [n_rows, dim] */
  const uint64_t *row_ids; /* [n_rows] or NULL = identity */
  int32_t device;
  uint32_t reserved;
} mi355_flat_desc;

/* analyze_plan-style counters (table/query.rs:105-112) for the last search */
typedef struct mi355_stats {
  uint32_t struct_size;
  uint32_t n_queries;
  uint64_t partitions_probed;  /* sum over queries */
  uint64_t vectors_scanned;    /* sum over (query, partition) pairs */
  uint64_t code_bytes_scanned; /* algorithmic bytes: m*nbits/8 per vector per query */
  uint64_t work_items;         /* scan work items launched; for batches cut by rows on the device (up to 512 (query, partition) pairs
                                  that cannot fill the chip) the candidate-slot groups laid out for them: an upper bound of the items */
  float us_coarse;             /* per-stage device time; 0 unless profiling on */
  float us_select;
  float us_plan;               /* between probe selection and the scan kernel: work list + table images */
  float us_scan;
  float us_merge;
  float us_refine;
  float us_total;
  uint32_t scan_variant;       /* which ADC kernel ran (MI355_SCAN_*) */
  uint32_t scan_launches;      /* timed launch sequences folded into us_* */
  uint32_t timed_out;          /* 1 = the device-side deadline of timeout_ms stopped the last call */
  uint32_t bad_probes;         /* mi355_search_probes: probe ids that are not partitions of this
                                  index (they are skipped; host-I/O calls also fail with InvalidInput) */
  uint32_t coalesced_calls;    /* concurrent host-I/O mi355_search calls served by the last device batch */
  uint32_t graph_replays;      /* searches of this handle served by a cached hipGraph so far */
  uint32_t lut_images;         /* 1 = the distance tables of the last search came from the batch-level table
                                  kernel (us_plan holds its time), 0 = every work item built its own */
} mi355_stats;

enum {
  MI355_SCAN_AUTO = 0,
  MI355_SCAN_PAIR = 1,   /* generic: one workgroup per (query, partition slice),
                            [sub-quantiser][code] table, any m */
  MI355_SCAN_SKEW = 2    /* production: pre-skewed code streams + [code][column]
                            table (bank-conflict-free gathers), partition-major
                            work queues per XCD; any m up to 768: a table holds
                            32 / 48 / 64 / 80 / 96 columns, other m <= 96 are padded
                            with zero columns, larger m are scanned in slabs of <= 96
                            columns (the row sum stays j-ascending); 4-bit codes are
                            expanded to one byte per sub-quantiser when packed */
};

/* mi355_index_configure `profile` bits above the low byte */
enum {
  MI355_PROFILE_MASK = 0xFFu,
  /* capture the launch sequence of small host-I/O batches (<= 64 queries) in a hipGraph,
     cached per (batch size, k, refine, nprobe, range / filter shape) */
  MI355_CFG_GRAPH = 0x100u,
  /* serve concurrent host-I/O mi355_search calls whose parameters agree from ONE device
     batch (leader / follower hand-off inside the handle, no extra thread) */
  MI355_CFG_COALESCE = 0x200u,
  /* opt-in: a device-I/O mi355_search with refine_factor over a HOST-resident raw column (mapped or
     attached) leaves its exact re-rank and final merge on a private stream of the handle, so that the
     NEXT call's scan overlaps the PCIe gather.  Contract of such a call: its outputs are NOT ordered
     on the handle's stream — they are complete after mi355_index_sync (or any other entry point of the
     handle, which all join the pending re-rank first); the caller keeps the query and output buffers
     untouched until then.  Off after open: without it every device-I/O result is ordered on the
     handle's stream. */
  MI355_CFG_DEFER_REFINE = 0x400u,
  /* opt-out: build every PQ distance table inside its scan work item.  By default an index whose sub-vectors are 16
     floats long (num_sub_vectors = dim / 16, the reference's default: rust/lancedb/src/index/vector.rs:306-319) gets the
     tables of a whole batch from one batch-level kernel that keeps the codebook in registers (k * refine_factor <= 128;
     up to 16 GiB of table images in HBM per batch chunk, falling back to in-item builds when HBM is full).  Results are
     bit-identical either way (same operations in the same order); the bit is for A/B measurements and tests. */
  MI355_CFG_LUT_INLINE = 0x800u
};

/* ---- library ------------------------------------------------------------ */
uint32_t mi355_abi_version(void);
/* number of visible gfx950 devices; 0 when none / no HIP runtime */
int32_t mi355_device_count(int32_t *out_count);
/* message of the last failing call on THIS thread (handle may be NULL) */
int32_t mi355_last_error(char *buf, size_t buf_len);

/* ---- IVF-PQ index lifecycle (replaces the lance Session index cache,
 *      python/src/session.rs:49-50: upload once, reuse across queries) ----- */
int32_t mi355_index_open(const mi355_index_desc *desc, mi355_index **out);
int32_t mi355_index_close(mi355_index *index);
/* run all later work of this handle on `hip_stream` (a hipStream_t); NULL
   restores the handle's own stream */
int32_t mi355_index_set_stream(mi355_index *index, void *hip_stream);
int32_t mi355_index_sync(mi355_index *index);
/* tuning knobs: MI355_SCAN_* variant (AUTO = the one the index layout was
   packed for at open; a mismatching explicit choice is INVALID_INPUT), slice
   length (rows per scan work item of the generic kernel, 0 = default) and
   profiling (low byte of `profile`): 0 = counters only, 1 = also per-stage device
   times of the LAST search, 2 = counters and times ACCUMULATE over searches
   until the next configure().  Times come from hipEvents recorded on the search
   stream with no host synchronisation; they are read back by mi355_last_stats.
   Higher bits: MI355_CFG_DEFER_REFINE (above), MI355_CFG_GRAPH / MI355_CFG_COALESCE (latency / concurrency modes,
   host-I/O calls only; after mi355_index_open coalescing is ON and graph replay is OFF —
   replay measured slower than eager launches; a configure() call sets
   both as given).  A configure() call also drops the handle's captured graphs. */
int32_t mi355_index_configure(mi355_index *index, uint32_t scan_variant,
                              uint32_t slice_rows, uint32_t profile);
/* Attach (or replace) the raw vector column of an open handle WITHOUT copying it: a DEVICE array
   [rows kept on this handle, dim] in the handle's local row order (= index order for an unsharded
   handle) that the caller keeps alive until detach / close.  For columns that should not exist
   twice in HBM (100 M x 768 bf16 = 154 GB).  Detach restores the column given at open, if any. */
int32_t mi355_index_attach_raw(mi355_index *index, const void *raw_vectors, uint32_t raw_dtype);
int32_t mi355_index_detach_raw(mi355_index *index);
/* rows kept on this handle and how many partitions are non-empty here */
int32_t mi355_index_info(const mi355_index *index, uint64_t *out_rows,
                         uint32_t *out_partitions_owned);

/*
 * IVF-PQ search — replaces ANNIvfPartitionExec + ANNIvfSubIndexExec + TopK
 * (+ refine Take/KNNVectorDistance).  `queries` is [n_queries, dim] f32
 * (Query::nearest_to always casts to Float32, query.rs:1011-1021).
 * Outputs: out_rowids/out_dist are [n_queries, k]; row q holds out_counts[q]
 * valid entries sorted by (distance, rowid); the tail is filled with
 * UINT64_MAX / +inf.
 */
int32_t mi355_search(mi355_index *index, const float *queries,
                     uint32_t n_queries, const mi355_search_params *params,
                     uint64_t *out_rowids, float *out_dist,
                     uint32_t *out_counts);
/*
 * Two-phase search for indexes whose coarse stage is worth sharding (C4: nlist =
 * 65536 -> 201 MB of centroids and 100 MFLOP per query; SURVEY.md §8e "sharded
 * coarse").  Phase 1: every rank scores a slice [cent_lo, cent_hi) of the centroids
 * and returns its best min(nprobe, slice) partitions as (partition id, coarse
 * distance) lists in mi355_merge_topk's layout; after an all-gather,
 * mi355_merge_topk(k = nprobe) orders them by (distance, partition id) = the probe
 * list of the unsharded search.  Phase 2: mi355_search_probes scans the partitions
 * of that list this handle owns (nprobe_min / nprobe_max of `params` are ignored).
 */
int32_t mi355_coarse_topn(mi355_index *index, const float *queries,
                          uint32_t n_queries, uint32_t nprobe, uint32_t cent_lo,
                          uint32_t cent_hi, uint32_t io_mem,
                          uint64_t *out_part_ids /*[n_q, nprobe]*/,
                          float *out_dist /*[n_q, nprobe]*/,
                          uint32_t *out_counts /*[n_q]*/);
int32_t mi355_search_probes(mi355_index *index, const float *queries,
                            uint32_t n_queries,
                            const mi355_search_params *params,
                            const uint64_t *probes /*[n_q, nprobe]*/,
                            uint32_t nprobe, uint64_t *out_rowids,
                            float *out_dist, uint32_t *out_counts);

/* waits for the handle's stream, then returns the counters (see profile modes) */
int32_t mi355_last_stats(mi355_index *index, mi355_stats *out);

/* ---- flat (no index / bypass_vector_index, query.rs:1360-1370) ---------- */
int32_t mi355_flat_open(const mi355_flat_desc *desc, mi355_flat **out);
int32_t mi355_flat_close(mi355_flat *flat);
int32_t mi355_flat_set_stream(mi355_flat *flat, void *hip_stream);
int32_t mi355_flat_sync(mi355_flat *flat);
/* Replaces KNNVectorDistance + SortExec TopK.  params->metric selects the
   metric (DEFAULT = L2, lib.rs:236-243); nprobe / refine fields are ignored. */
int32_t mi355_flat_search(mi355_flat *flat, const float *queries,
                          uint32_t n_queries,
                          const mi355_search_params *params,
                          uint64_t *out_rowids, float *out_dist,
                          uint32_t *out_counts);

/* Tuning of the flat GEMM filter.  gemm_variant: MI355_FLAT_GEMM_* (AUTO = the library's
   choice per batch size).  grid_workgroups: 1 = one workgroup per tile, N >= 8 = a persistent
   grid of N / 8 * 8 workgroups (each walks its XCD's tiles), 0 = the library's choice for the
   schedule.  flags: MI355_FLAT_CHECKSUM = after every GEMM launch synchronise and keep an
   order-independent checksum + census of the group-minimum matrix (mi355_flat_checksum /
   mi355_flat_census; used to compare schedules bit for bit). */
enum {
  MI355_FLAT_GEMM_AUTO = 0,
  MI355_FLAT_GEMM_128 = 1,       /* 128 x 128 tile, 4 waves, two barriers per k-step */
  MI355_FLAT_GEMM_256 = 2,       /* 256 x 256 tile, 8 waves, two barriers per k-step */
  /* 3, 6, 7, 8: schedules measured as no gain in rounds 1-2 and removed (profiles/r02_*) */
  MI355_FLAT_GEMM_8PHASE = 4,    /* 256 x 256, 8-phase schedule (counted vmcnt, staggered wave
                                    groups), fast epilogue: AUTO's choice above 128 queries */
  MI355_FLAT_GEMM_8PHASE_REF = 5 /* the same schedule with variant 2's epilogue arithmetic
                                    (bit-identical filter matrix: the schedule's own check) */
};
enum {
  MI355_FLAT_CHECKSUM = 1u,
  /* accumulate the GEMM kernel's own device time (HIP events recorded around its launches on the
     search stream, no host synchronisation) until the next configure(): mi355_flat_last_stats */
  MI355_FLAT_PROFILE = 2u,
  /* Path choice.  By default a search takes the cheaper of the two exact paths by a cost model of the
     call (rows x queries): the bf16 MFMA filter + exact re-rank pays ~0.3 ms of fixed launches and pads
     the batch to whole 256-query tiles, the exact sweep re-reads the column once per query — a single
     query, or a table of a few hundred thousand rows, is faster swept.  Both return the same bits.
     These two pin the choice (tests and A/B runs; FILTER is ignored when the call cannot be filtered:
     lower bound, prefilter, a column opened without filter data). */
  MI355_FLAT_FORCE_FILTER = 4u,
  MI355_FLAT_FORCE_SWEEP = 8u
};
int32_t mi355_flat_configure(mi355_flat *flat, uint32_t gemm_variant,
                             uint32_t grid_workgroups, uint32_t flags);
int32_t mi355_flat_checksum(mi355_flat *flat, uint64_t *out_checksum);
/* census of the same matrix (MI355_FLAT_CHECKSUM searches): entries marked "never filter" (-inf),
   entries that are NaN / +inf (must be 0) and the sum of the finite ones */
int32_t mi355_flat_census(mi355_flat *flat, uint64_t *out_never_filter, uint64_t *out_not_finite, double *out_sum);

typedef struct mi355_flat_stats {
  uint32_t struct_size;
  uint32_t gemm_variant;   /* MI355_FLAT_GEMM_* the last search ran */
  uint32_t gemm_launches;  /* GEMM launches folded into us_gemm */
  uint32_t fallback_queries; /* queries of the LAST search whose candidate list overflowed and were swept
                                exactly (adversarial columns; 0 on ordinary data — a non-zero value on
                                random data means the filter GEMM produced garbage) */
  float us_gemm;           /* summed device time of those launches */
  float us_rest;           /* query prep + thresholds + compaction + exact re-rank of the same searches */
  uint64_t gemm_flops;     /* algorithmic flops of those launches: 2 * queries (padded) * rows * dim (padded) */
} mi355_flat_stats;
/* waits for the handle's stream, then returns the accumulated numbers */
int32_t mi355_flat_last_stats(mi355_flat *flat, mi355_flat_stats *out);

/* which kernels served the last mi355_flat_search: 1 = bf16 MFMA GEMM filter +
   exact re-rank, 2 = exact scalar sweep (small columns, lower-bounded ranges);
   out_has_filter = 1 when the handle carries the filter data (bf16 shadow / row
   norms, built at open for columns of >= 4096 rows). */
int32_t mi355_flat_info(const mi355_flat *flat, uint32_t *out_last_path,
                        uint32_t *out_has_filter);

/*
 * Merge n_lists candidate lists per query into one top-k (the reducer after
 * the multi-GPU all-gather).  Inputs are DEVICE pointers laid
 * out [n_lists, n_queries, k] (+ counts [n_lists, n_queries]); outputs are
 * DEVICE [n_queries, k] / [n_queries].  Runs on `hip_stream` of `device`.
 */
int32_t mi355_merge_topk(int32_t device, void *hip_stream,
                         const uint64_t *in_rowids, const float *in_dist,
                         const uint32_t *in_counts, uint32_t n_lists,
                         uint32_t n_queries, uint32_t k, uint64_t *out_rowids,
                         float *out_dist, uint32_t *out_counts);

/*
 * Index population (first half): the O(N) transform stage of
 * the reference's index build — given TRAINED IVF centroids and a PQ codebook
 * (IvfBuildParams / PQBuildParams, rust/lancedb/src/table/create_index.rs:283-303),
 * assign every row to its partition, PQ-encode its residual and lay the rows out
 * partition by partition: exactly the arrays mi355_index_open takes
 * (MI355_CODES_ROW_MAJOR).  k-means training itself runs on samples and is not part
 * of this entry point.  Definitions (oracle/ann_oracle.c orc_ivfpq_encode):
 *   cosine: rows are normalised first (x / sqrt(chain_dot(x,x)))
 *   partition(x) = argmin_p coarse(x, c_p), ties to the lower p (the search's coarse formula)
 *   code_j(x)    = argmin_c chain_l2(r_j, codebook[j][c]) with r = x - c_partition
 *                  (dot: r = x, argmin_c 1 - chain_dot(x_j, codebook[j][c])), ties to the lower c
 *   order        = rows sorted by (partition, source row): stable
 */
typedef struct mi355_encode_desc {
  uint32_t struct_size;
  uint32_t dim, nlist, m, nbits, metric;
  uint32_t mem;       /* MI355_MEM_*: where vectors and the outputs live */
  int32_t device;
  const float *centroids; /* [nlist, dim], same memory as `mem` */
  const float *codebook;  /* [m, 2^nbits, dim/m] */
} mi355_encode_desc;

int32_t mi355_ivfpq_encode(const mi355_encode_desc *desc, const float *vectors /*[n_rows, dim] f32*/,
                           uint64_t n_rows, uint64_t *out_part_offsets /*[nlist+1], ALWAYS host*/,
                           uint8_t *out_codes /*[n_rows, m * nbits / 8], index order
                                                (4-bit: sub-quantiser 2t in the low nibble of byte t)*/,
                           uint64_t *out_order /*[n_rows]: source row at each index position*/,
                           uint32_t *out_assign /*[n_rows] partition of each SOURCE row, or NULL*/);

/*
 * Index training (second half): the k-means behind
 * IvfBuildParams (num_partitions, max_iterations, sample_rate) and PQBuildParams
 * (one 256-entry codebook per sub-vector, trained on residuals) —
 * rust/lancedb/src/index/vector.rs:61-119, create_index.rs:68-102, :283-303; the
 * reference already sends this stage out of process for accelerators
 * (python/python/lancedb/table.py:2883-2937).  lance's own trainer [EXT] draws a random
 * initialisation, so centroid-for-centroid parity with it is undefined; this entry point
 * is a DETERMINISTIC Lloyd iteration — the caller supplies the initial centroids — whose
 * result is bit-exact against oracle/ann_oracle.c orc_kmeans_train:
 *   cosine: rows are normalised first (as mi355_ivfpq_encode)
 *   repeat `iters` times:
 *     assign(x) = argmin_c coarse(x, c), ties to the lower c   (the search's coarse formula)
 *     c <- (sum of its rows, added one by one in source-row order, f32) / count  (IEEE divide);
 *          a centroid without rows keeps its value
 * `ld` is the distance in floats between consecutive rows (0 = dim): a PQ sub-quantiser is
 * trained on columns [j*dsub, (j+1)*dsub) of the residual matrix without copying it out.
 */
typedef struct mi355_kmeans_desc {
  uint32_t struct_size;
  uint32_t dim, k, metric, iters;
  uint32_t mem;   /* MI355_MEM_*: vectors, centroids and the outputs */
  int32_t device;
  uint32_t reserved0;
  uint64_t ld;    /* row stride of `vectors` in floats; 0 = dim */
} mi355_kmeans_desc;

int32_t mi355_kmeans_train(const mi355_kmeans_desc *desc, const float *vectors, uint64_t n_rows,
                           float *centroids /*[k, dim] in: initial, out: trained*/,
                           uint64_t *out_counts /*[k] rows per centroid at the last assignment, or NULL*/);

/* All m PQ sub-quantisers of PQBuildParams in ONE call: sub-quantiser j is exactly
   mi355_kmeans_train (k = 2^nbits, plain L2 — dot for dot indexes — `iters` Lloyd iterations) on
   columns [j * dim/m, (j+1) * dim/m) of the residual matrix, but the matrix crosses to the device
   once and the m trainers share one stream and one scratch set. */
typedef struct mi355_pq_train_desc {
  uint32_t struct_size;
  uint32_t dim, m, nbits;
  uint32_t metric;  /* the INDEX metric (MI355_METRIC_*) */
  uint32_t iters;
  uint32_t mem;     /* MI355_MEM_*: residuals and codebook */
  int32_t device;
} mi355_pq_train_desc;
int32_t mi355_pq_train(const mi355_pq_train_desc *desc, const float *residuals /*[n_rows, dim]*/, uint64_t n_rows,
                       float *codebook /*[m, 2^nbits, dim/m] in: initial entries, out: trained*/);

/* Residuals of the rows to their partition centroid, the training set of the PQ
   codebooks: out[i] = x_i - c[assign(x_i)] (cosine: x normalised first; dot: out = x).
   out_assign may be NULL. */
int32_t mi355_ivf_residuals(const mi355_kmeans_desc *desc /*k = nlist; iters, ld as above*/,
                            const float *vectors, uint64_t n_rows, const float *centroids,
                            float *out_residuals /*[n_rows, dim] dense*/, uint32_t *out_assign);

/*
 * ---- multi-GPU exchange behind the ABI -------------------------
 * The reference has no collective; this exchange is the engine's own:
 * one process per GPU, the IVF partition list sharded across ranks
 * (mi355_index_desc.shard_count / shard_rank = world / rank), RCCL over xGMI for the
 * candidate all-gather.  No PyTorch: a Rust host creates the communicator from a
 * 128-byte id it distributes over its own channel (file, socket, MPI ...).
 */
typedef struct mi355_comm mi355_comm;
#define MI355_COMM_ID_BYTES 128
/* rank 0: a fresh id (ncclGetUniqueId) */
int32_t mi355_comm_unique_id(void *out_id /*[MI355_COMM_ID_BYTES]*/);
/* every rank, collectively (ncclCommInitRank on `device`); world = 1 needs no peers.
   librccl is loaded (dlopen) by the first mi355_comm_unique_id / mi355_comm_create call: a host
   that never shards needs no RCCL installed. */
int32_t mi355_comm_create(const void *id, uint32_t rank, uint32_t world, int32_t device,
                          mi355_comm **out);
/* Loopback transport: `world` communicators (out[0 .. world)) whose ranks all live in THIS
   process on ONE device.  Rank r's collective calls must come from its own thread with shard
   handle r of `world` — the gather is a host rendezvous plus device copies between the ranks'
   slabs, stream-ordered by events; everything above the gather (slab layout, merge with owners,
   owner-side refine, the collective second pass, the sharded coarse stage) is the code the RCCL
   transport runs.  Purpose: execute and time the world > 1 path on a single GPU.  A rank that
   fails inside a collective aborts the group (its peers return MI355_ERR_RUNTIME instead of
   waiting); each communicator is released with mi355_comm_destroy. */
int32_t mi355_comm_create_loopback(uint32_t world, int32_t device, mi355_comm **out /*[world]*/);
int32_t mi355_comm_destroy(mi355_comm *comm);

enum {
  /* score only this rank's slice of the centroids and select the probe list after an
     all-gather of per-rank (partition, distance) pairs (C4: nlist = 65536) */
  MI355_SHARD_COARSE = 1u,
  /* keep the exchange on the handle's stream.  By default a device-I/O call without a timeout
     and without maximum_nprobes expansion queues its exchange (gather, merge, owner-side refine,
     second gather, final merge) on the communicator's own stream, so the NEXT call's scan
     overlaps it; its outputs are complete after mi355_index_sync, and
     every other entry point of the handle joins the exchange first. */
  MI355_SHARD_NO_OVERLAP = 2u
};

/* per-rank load of the last mi355_search_sharded call, identical on every rank */
#define MI355_MAX_RANKS 64
typedef struct mi355_comm_stats {
  uint32_t struct_size;
  uint32_t world;
  uint32_t rank;
  uint32_t n_gathers;                   /* RCCL all-gathers issued by the last call */
  uint64_t bytes_gathered;              /* received bytes, this rank */
  uint64_t rows_scanned[MI355_MAX_RANKS]; /* ADC rows per rank (sum over its work items) */
  float imbalance;                      /* max / mean of rows_scanned */
  uint32_t reserved;
  float us_exchange;                    /* device time of the last call's exchange: first gather .. final merge */
  uint32_t overlapped;                  /* 1: that exchange ran on the communicator's stream (MI355_SHARD_NO_OVERLAP unset) */
} mi355_comm_stats;
int32_t mi355_comm_last_stats(mi355_comm *comm, mi355_comm_stats *out);

/*
 * The sharded search: called collectively by every rank with the SAME queries and
 * params on its own shard handle; the result is identical on every rank and
 * identical to the unsharded mi355_search — including refine_factor (the global
 * k * refine_factor ANN candidates are selected FIRST, each rank then refines the
 * candidates whose raw vectors it owns and a second all-gather + merge keeps k:
 * rank-sharded raw vectors, C5) and maximum_nprobes (the expansion decision is taken
 * on the merged counts, so all ranks take it together).
 * One packed all-gather per exchange ([B, kk] x 16-B candidate records + [B] counts),
 * then k_merge on every rank.  Device-I/O calls return without waiting and never
 * synchronise with the host — the short queries of maximum_nprobes are picked on the
 * device; queries / outputs live where params->io_mem says and must stay untouched until
 * mi355_index_sync (see MI355_SHARD_NO_OVERLAP for the stream the exchange runs on).
 * With a timeout the status is the OR of every rank's device-side deadline flag (it
 * travels in the slab trailer), so every rank returns the same status.
 */
int32_t mi355_search_sharded(mi355_index *index, mi355_comm *comm, const float *queries,
                             uint32_t n_queries, const mi355_search_params *params,
                             uint32_t flags, uint64_t *out_rowids, float *out_dist,
                             uint32_t *out_counts);

/* Flat search with the rows sharded across ranks (SURVEY.md §8e "replicas-only
   fallback: shard rows, same final all-gather"): every rank searches its own slice of
   the column (row_ids carry the global ids), then the same gather + merge. */
int32_t mi355_flat_search_sharded(mi355_flat *flat, mi355_comm *comm, const float *queries,
                                  uint32_t n_queries, const mi355_search_params *params,
                                  uint64_t *out_rowids, float *out_dist,
                                  uint32_t *out_counts);

/* centroid slice [lo, hi) rank scores in the sharded coarse stage (contiguous, balanced) */
int32_t mi355_coarse_slice(uint32_t nlist, uint32_t world, uint32_t rank, uint32_t *out_lo,
                           uint32_t *out_hi);

/* Deterministic partition -> shard assignment used by mi355_index_open
   (greedy: partitions by descending length, each to the least loaded shard;
   ties to the lower shard id).  out_owner is [nlist]. */
int32_t mi355_shard_plan(const uint64_t *part_offsets, uint32_t nlist,
                         uint32_t shard_count, uint32_t *out_owner);
/* The same greedy packing over cost[p] = rows(p) * weight[p] (ties: longer partition, then lower id): with
   weight = how often partition p is probed (a histogram of mi355_coarse_topn over a sample of the query load)
   the shards are balanced by the bytes they SCAN, not by the bytes they hold — equal rows owned still left
   +- 8 % of scanned rows between eight shards of the C3 bench index, whose popular partitions are probed
   several times as often as the others.  weight NULL = all 1 (= mi355_shard_plan).  Hand the result to every
   rank's mi355_index_desc.part_owner. */
int32_t mi355_shard_plan_weighted(const uint64_t *part_offsets, const float *weight, uint32_t nlist,
                                  uint32_t shard_count, uint32_t *out_owner);

#ifdef __cplusplus
}
#endif
#endif /* MI355_ANN_H */
