"""ctypes mirror of include/mi355_ann.h (struct layouts and constants only).

This is the Python-side analogue of the binding a maintainer would add on the
Rust side: field order and widths must match the header
exactly; `struct_size` guards against drift at run time.
"""
import ctypes as C

ABI_VERSION = 5

OK = 0
ERR_INVALID_INPUT = 1
ERR_RUNTIME = 2
ERR_TIMEOUT = 3
ERR_NOT_SUPPORTED = 4

METRIC_L2 = 0
METRIC_COSINE = 1
METRIC_DOT = 2
METRIC_DEFAULT = 255

MEM_HOST = 0
MEM_DEVICE = 1

DTYPE_F32 = 0
DTYPE_BF16 = 1
DTYPE_F16 = 2

CODES_ROW_MAJOR = 0
CODES_PART_TRANSPOSED = 1

SCAN_AUTO = 0
SCAN_PAIR = 1
SCAN_SKEW = 2

FILTER_NONE = 0
FILTER_ALLOW = 1
FILTER_BLOCK = 2

APPROX_UNSET = 0
APPROX_FAST = 1
APPROX_NORMAL = 2
APPROX_ACCURATE = 3
APPROX_NAMES = {"fast": APPROX_FAST, "normal": APPROX_NORMAL, "accurate": APPROX_ACCURATE}

INDEX_GENERIC_SCAN = 1
INDEX_RAW_HOST_MAPPED = 2
INDEX_LOCAL_ARRAYS = 4

PROFILE_MASK = 0xFF
CFG_GRAPH = 0x100
CFG_COALESCE = 0x200
CFG_DEFER_REFINE = 0x400
CFG_LUT_INLINE = 0x800

FLAT_GEMM_AUTO = 0
FLAT_GEMM_128 = 1
FLAT_GEMM_256 = 2
FLAT_GEMM_8PHASE = 4
FLAT_GEMM_8PHASE_REF = 5
FLAT_CHECKSUM = 1
FLAT_PROFILE = 2
FLAT_FORCE_FILTER = 4
FLAT_FORCE_SWEEP = 8

SHARD_COARSE = 1
SHARD_NO_OVERLAP = 2
COMM_ID_BYTES = 128
MAX_RANKS = 64

UINT64_MAX = 0xFFFFFFFFFFFFFFFF


class IndexDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("dim", C.c_uint32),
        ("nlist", C.c_uint32),
        ("m", C.c_uint32),
        ("nbits", C.c_uint32),
        ("metric", C.c_uint32),
        ("n_rows", C.c_uint64),
        ("mem", C.c_uint32),
        ("codes_layout", C.c_uint32),
        ("centroids", C.c_void_p),
        ("codebook", C.c_void_p),
        ("part_offsets", C.c_void_p),
        ("codes", C.c_void_p),
        ("row_ids", C.c_void_p),
        ("raw_vectors", C.c_void_p),
        ("raw_dtype", C.c_uint32),
        ("device", C.c_int32),
        ("shard_count", C.c_uint32),
        ("shard_rank", C.c_uint32),
        ("flags", C.c_uint32),
        ("reserved", C.c_uint32),
        ("part_owner", C.c_void_p),
    ]


class SearchParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("k", C.c_uint32),
        ("nprobe_min", C.c_uint32),
        ("nprobe_max", C.c_uint32),
        ("refine_factor", C.c_uint32),
        ("metric", C.c_uint32),
        ("has_lower_bound", C.c_uint32),
        ("has_upper_bound", C.c_uint32),
        ("lower_bound", C.c_float),
        ("upper_bound", C.c_float),
        ("io_mem", C.c_uint32),
        ("timeout_ms", C.c_uint32),
        ("filter_mode", C.c_uint32),
        ("approx_mode", C.c_uint32),
        ("filter_rowids", C.c_void_p),
        ("n_filter", C.c_uint64),
    ]


class FlatDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("dim", C.c_uint32),
        ("n_rows", C.c_uint64),
        ("dtype", C.c_uint32),
        ("mem", C.c_uint32),
        ("vectors", C.c_void_p),
        ("row_ids", C.c_void_p),
        ("device", C.c_int32),
        ("reserved", C.c_uint32),
    ]


class EncodeDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("dim", C.c_uint32),
        ("nlist", C.c_uint32),
        ("m", C.c_uint32),
        ("nbits", C.c_uint32),
        ("metric", C.c_uint32),
        ("mem", C.c_uint32),
        ("device", C.c_int32),
        ("centroids", C.c_void_p),
        ("codebook", C.c_void_p),
    ]


class KmeansDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("dim", C.c_uint32),
        ("k", C.c_uint32),
        ("metric", C.c_uint32),
        ("iters", C.c_uint32),
        ("mem", C.c_uint32),
        ("device", C.c_int32),
        ("reserved0", C.c_uint32),
        ("ld", C.c_uint64),
    ]


class PqTrainDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("dim", C.c_uint32),
        ("m", C.c_uint32),
        ("nbits", C.c_uint32),
        ("metric", C.c_uint32),
        ("iters", C.c_uint32),
        ("mem", C.c_uint32),
        ("device", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n_queries", C.c_uint32),
        ("partitions_probed", C.c_uint64),
        ("vectors_scanned", C.c_uint64),
        ("code_bytes_scanned", C.c_uint64),
        ("work_items", C.c_uint64),
        ("us_coarse", C.c_float),
        ("us_select", C.c_float),
        ("us_plan", C.c_float),
        ("us_scan", C.c_float),
        ("us_merge", C.c_float),
        ("us_refine", C.c_float),
        ("us_total", C.c_float),
        ("scan_variant", C.c_uint32),
        ("scan_launches", C.c_uint32),
        ("timed_out", C.c_uint32),
        ("bad_probes", C.c_uint32),
        ("coalesced_calls", C.c_uint32),
        ("graph_replays", C.c_uint32),
        ("lut_images", C.c_uint32),
    ]


class FlatStats(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("gemm_variant", C.c_uint32),
        ("gemm_launches", C.c_uint32),
        ("fallback_queries", C.c_uint32),
        ("us_gemm", C.c_float),
        ("us_rest", C.c_float),
        ("gemm_flops", C.c_uint64),
    ]


class CommStats(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("world", C.c_uint32),
        ("rank", C.c_uint32),
        ("n_gathers", C.c_uint32),
        ("bytes_gathered", C.c_uint64),
        ("rows_scanned", C.c_uint64 * MAX_RANKS),
        ("imbalance", C.c_float),
        ("reserved", C.c_uint32),
        ("us_exchange", C.c_float),
        ("overlapped", C.c_uint32),
    ]


# every symbol include/mi355_ann.h declares (tests check the .so exports all)
EXPORTED_SYMBOLS = (
    "mi355_abi_version",
    "mi355_device_count",
    "mi355_last_error",
    "mi355_index_open",
    "mi355_index_close",
    "mi355_index_set_stream",
    "mi355_index_sync",
    "mi355_index_configure",
    "mi355_index_info",
    "mi355_index_attach_raw",
    "mi355_index_detach_raw",
    "mi355_search",
    "mi355_coarse_topn",
    "mi355_search_probes",
    "mi355_last_stats",
    "mi355_flat_open",
    "mi355_flat_close",
    "mi355_flat_set_stream",
    "mi355_flat_sync",
    "mi355_flat_search",
    "mi355_flat_info",
    "mi355_flat_configure",
    "mi355_flat_checksum",
    "mi355_flat_census",
    "mi355_flat_last_stats",
    "mi355_comm_unique_id",
    "mi355_comm_create",
    "mi355_comm_create_loopback",
    "mi355_comm_destroy",
    "mi355_comm_last_stats",
    "mi355_search_sharded",
    "mi355_flat_search_sharded",
    "mi355_coarse_slice",
    "mi355_ivfpq_encode", "mi355_kmeans_train", "mi355_ivf_residuals", "mi355_pq_train",
    "mi355_merge_topk",
    "mi355_shard_plan",
    "mi355_shard_plan_weighted",
)

METRIC_NAMES = {"l2": METRIC_L2, "cosine": METRIC_COSINE, "dot": METRIC_DOT}


def make_params(k=10, nprobe_min=20, nprobe_max=20, refine_factor=0,
                metric=METRIC_DEFAULT, lower_bound=None, upper_bound=None,
                io_mem=MEM_HOST, timeout_ms=0, allow_rowids=None, block_rowids=None, approx_mode=None):
    """Fill a SearchParams with VectorQueryRequest defaults
    (rust/lancedb/src/query.rs:1097-1114: nprobes 20/20, k 10, no refine)."""
    p = SearchParams()
    p.struct_size = C.sizeof(SearchParams)
    p.k = k
    p.nprobe_min = nprobe_min
    p.nprobe_max = 0 if nprobe_max is None else nprobe_max
    p.refine_factor = refine_factor or 0
    p.metric = metric
    p.has_lower_bound = 0 if lower_bound is None else 1
    p.has_upper_bound = 0 if upper_bound is None else 1
    p.lower_bound = 0.0 if lower_bound is None else float(lower_bound)
    p.upper_bound = 0.0 if upper_bound is None else float(upper_bound)
    p.io_mem = io_mem
    p.timeout_ms = timeout_ms
    if approx_mode is None:
        p.approx_mode = APPROX_UNSET
    elif isinstance(approx_mode, str):  # lib.rs:343-357
        if approx_mode.lower() not in APPROX_NAMES:
            raise ValueError(f"approx_mode must be one of 'fast', 'normal', or 'accurate', got '{approx_mode}'")
        p.approx_mode = APPROX_NAMES[approx_mode.lower()]
    else:
        p.approx_mode = int(approx_mode)
    # prefilter: a sorted, unique u64 array (numpy for host I/O, a device array for device I/O);
    # the array is kept alive on the params object
    flt = allow_rowids if allow_rowids is not None else block_rowids
    p.filter_mode = FILTER_NONE
    if flt is not None:
        if allow_rowids is not None and block_rowids is not None:
            raise ValueError("give either allow_rowids or block_rowids")
        if hasattr(flt, "data_ptr"):
            p._keep, ptr, n = flt, flt.data_ptr(), int(flt.shape[0])
        else:
            import numpy as np
            arr = np.unique(np.ascontiguousarray(flt, dtype=np.uint64))
            p._keep, ptr, n = arr, arr.ctypes.data, int(arr.size)
        p.filter_mode = FILTER_ALLOW if allow_rowids is not None else FILTER_BLOCK
        p.filter_rowids = ptr
        p.n_filter = n
    return p
