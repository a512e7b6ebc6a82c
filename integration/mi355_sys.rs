// mi355_sys.rs — GENERATED from include/mi355_ann.h by scripts/gen_rust_sys.py; do not edit by hand.
// Raw `extern "C"` bindings of libmi355_ann.so for the Rust shim of INTEGRATION.md (the `BaseTable` /
// `ExecutionPlan` side of rust/lancedb/src/table.rs:549-576 and table/query.rs:131-328).  Never compiled in this
// repository (no Rust toolchain in the image); tests/test_rust_binding.py checks it against the header instead:
// identical function set, argument counts, struct field order, field offsets and sizes.
#![allow(non_camel_case_types, non_upper_case_globals, dead_code)]
use core::ffi::{c_char, c_void};

// ---- constants (48)
pub const MI355_ANN_ABI_VERSION: u32 = 5;
pub const MI355_COMM_ID_BYTES: usize = 128;
pub const MI355_MAX_RANKS: usize = 64;
pub const MI355_OK: i32 = 0;
pub const MI355_ERR_INVALID_INPUT: i32 = 1;
pub const MI355_ERR_RUNTIME: i32 = 2;
pub const MI355_ERR_TIMEOUT: i32 = 3;
pub const MI355_ERR_NOT_SUPPORTED: i32 = 4;
pub const MI355_METRIC_L2: u32 = 0;
pub const MI355_METRIC_COSINE: u32 = 1;
pub const MI355_METRIC_DOT: u32 = 2;
pub const MI355_METRIC_DEFAULT: u32 = 255;
pub const MI355_MEM_HOST: u32 = 0;
pub const MI355_MEM_DEVICE: u32 = 1;
pub const MI355_DTYPE_F32: u32 = 0;
pub const MI355_DTYPE_BF16: u32 = 1;
pub const MI355_DTYPE_F16: u32 = 2;
pub const MI355_APPROX_UNSET: u32 = 0;
pub const MI355_APPROX_FAST: u32 = 1;
pub const MI355_APPROX_NORMAL: u32 = 2;
pub const MI355_APPROX_ACCURATE: u32 = 3;
pub const MI355_INDEX_GENERIC_SCAN: u32 = 1;
pub const MI355_INDEX_RAW_HOST_MAPPED: u32 = 2;
pub const MI355_INDEX_LOCAL_ARRAYS: u32 = 4;
pub const MI355_CODES_ROW_MAJOR: u32 = 0;
pub const MI355_CODES_PART_TRANSPOSED: u32 = 1;
pub const MI355_FILTER_NONE: u32 = 0;
pub const MI355_FILTER_ALLOW: u32 = 1;
pub const MI355_FILTER_BLOCK: u32 = 2;
pub const MI355_SCAN_AUTO: u32 = 0;
pub const MI355_SCAN_PAIR: u32 = 1;
pub const MI355_SCAN_SKEW: u32 = 2;
pub const MI355_PROFILE_MASK: u32 = 255;
pub const MI355_CFG_GRAPH: u32 = 256;
pub const MI355_CFG_COALESCE: u32 = 512;
pub const MI355_CFG_DEFER_REFINE: u32 = 1024;
pub const MI355_CFG_LUT_INLINE: u32 = 2048;
pub const MI355_FLAT_GEMM_AUTO: u32 = 0;
pub const MI355_FLAT_GEMM_128: u32 = 1;
pub const MI355_FLAT_GEMM_256: u32 = 2;
pub const MI355_FLAT_GEMM_8PHASE: u32 = 4;
pub const MI355_FLAT_GEMM_8PHASE_REF: u32 = 5;
pub const MI355_FLAT_CHECKSUM: u32 = 1;
pub const MI355_FLAT_PROFILE: u32 = 2;
pub const MI355_FLAT_FORCE_FILTER: u32 = 4;
pub const MI355_FLAT_FORCE_SWEEP: u32 = 8;
pub const MI355_SHARD_COARSE: u32 = 1;
pub const MI355_SHARD_NO_OVERLAP: u32 = 2;

// ---- opaque handles
#[repr(C)]
pub struct mi355_index {
    _private: [u8; 0],
}
#[repr(C)]
pub struct mi355_flat {
    _private: [u8; 0],
}
#[repr(C)]
pub struct mi355_comm {
    _private: [u8; 0],
}

// ---- descriptors and statistics (plain old data, `struct_size` = size_of::<Self>() as u32)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct mi355_index_desc {
    pub struct_size: u32,
    pub dim: u32,
    pub nlist: u32,
    pub m: u32,
    pub nbits: u32,
    pub metric: u32,
    pub n_rows: u64,
    pub mem: u32,
    pub codes_layout: u32,
    pub centroids: *const f32,
    pub codebook: *const f32,
    pub part_offsets: *const u64,
    pub codes: *const u8,
    pub row_ids: *const u64,
    pub raw_vectors: *const c_void,
    pub raw_dtype: u32,
    pub device: i32,
    pub shard_count: u32,
    pub shard_rank: u32,
    pub flags: u32,
    pub reserved: u32,
    pub part_owner: *const u32,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct mi355_search_params {
    pub struct_size: u32,
    pub k: u32,
    pub nprobe_min: u32,
    pub nprobe_max: u32,
    pub refine_factor: u32,
    pub metric: u32,
    pub has_lower_bound: u32,
    pub has_upper_bound: u32,
    pub lower_bound: f32,
    pub upper_bound: f32,
    pub io_mem: u32,
    pub timeout_ms: u32,
    pub filter_mode: u32,
    pub approx_mode: u32,
    pub filter_rowids: *const u64,
    pub n_filter: u64,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct mi355_flat_desc {
    pub struct_size: u32,
    pub dim: u32,
    pub n_rows: u64,
    pub dtype: u32,
    pub mem: u32,
    pub vectors: *const c_void,
    pub row_ids: *const u64,
    pub device: i32,
    pub reserved: u32,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct mi355_stats {
    pub struct_size: u32,
    pub n_queries: u32,
    pub partitions_probed: u64,
    pub vectors_scanned: u64,
    pub code_bytes_scanned: u64,
    pub work_items: u64,
    pub us_coarse: f32,
    pub us_select: f32,
    pub us_plan: f32,
    pub us_scan: f32,
    pub us_merge: f32,
    pub us_refine: f32,
    pub us_total: f32,
    pub scan_variant: u32,
    pub scan_launches: u32,
    pub timed_out: u32,
    pub bad_probes: u32,
    pub coalesced_calls: u32,
    pub graph_replays: u32,
    pub lut_images: u32,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct mi355_flat_stats {
    pub struct_size: u32,
    pub gemm_variant: u32,
    pub gemm_launches: u32,
    pub fallback_queries: u32,
    pub us_gemm: f32,
    pub us_rest: f32,
    pub gemm_flops: u64,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct mi355_encode_desc {
    pub struct_size: u32,
    pub dim: u32,
    pub nlist: u32,
    pub m: u32,
    pub nbits: u32,
    pub metric: u32,
    pub mem: u32,
    pub device: i32,
    pub centroids: *const f32,
    pub codebook: *const f32,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct mi355_kmeans_desc {
    pub struct_size: u32,
    pub dim: u32,
    pub k: u32,
    pub metric: u32,
    pub iters: u32,
    pub mem: u32,
    pub device: i32,
    pub reserved0: u32,
    pub ld: u64,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct mi355_pq_train_desc {
    pub struct_size: u32,
    pub dim: u32,
    pub m: u32,
    pub nbits: u32,
    pub metric: u32,
    pub iters: u32,
    pub mem: u32,
    pub device: i32,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct mi355_comm_stats {
    pub struct_size: u32,
    pub world: u32,
    pub rank: u32,
    pub n_gathers: u32,
    pub bytes_gathered: u64,
    pub rows_scanned: [u64; MI355_MAX_RANKS],
    pub imbalance: f32,
    pub reserved: u32,
    pub us_exchange: f32,
    pub overlapped: u32,
}

// ---- entry points (40); every status is 0 = ok, 1 InvalidInput, 2 Runtime, 3 Timeout, 4 NotSupported
// (rust/lancedb/src/error.rs:55-145); the message of a failing call: mi355_last_error on the same thread
#[link(name = "mi355_ann")]
extern "C" {
    pub fn mi355_abi_version() -> u32;
    pub fn mi355_device_count(out_count: *mut i32) -> i32;
    pub fn mi355_last_error(buf: *mut c_char, buf_len: usize) -> i32;
    pub fn mi355_index_open(desc: *const mi355_index_desc, out: *mut *mut mi355_index) -> i32;
    pub fn mi355_index_close(index: *mut mi355_index) -> i32;
    pub fn mi355_index_set_stream(index: *mut mi355_index, hip_stream: *mut c_void) -> i32;
    pub fn mi355_index_sync(index: *mut mi355_index) -> i32;
    pub fn mi355_index_configure(index: *mut mi355_index, scan_variant: u32, slice_rows: u32, profile: u32) -> i32;
    pub fn mi355_index_attach_raw(index: *mut mi355_index, raw_vectors: *const c_void, raw_dtype: u32) -> i32;
    pub fn mi355_index_detach_raw(index: *mut mi355_index) -> i32;
    pub fn mi355_index_info(index: *const mi355_index, out_rows: *mut u64, out_partitions_owned: *mut u32) -> i32;
    pub fn mi355_search(index: *mut mi355_index, queries: *const f32, n_queries: u32, params: *const mi355_search_params, out_rowids: *mut u64, out_dist: *mut f32, out_counts: *mut u32) -> i32;
    pub fn mi355_coarse_topn(index: *mut mi355_index, queries: *const f32, n_queries: u32, nprobe: u32, cent_lo: u32, cent_hi: u32, io_mem: u32, out_part_ids: *mut u64, out_dist: *mut f32, out_counts: *mut u32) -> i32;
    pub fn mi355_search_probes(index: *mut mi355_index, queries: *const f32, n_queries: u32, params: *const mi355_search_params, probes: *const u64, nprobe: u32, out_rowids: *mut u64, out_dist: *mut f32, out_counts: *mut u32) -> i32;
    pub fn mi355_last_stats(index: *mut mi355_index, out: *mut mi355_stats) -> i32;
    pub fn mi355_flat_open(desc: *const mi355_flat_desc, out: *mut *mut mi355_flat) -> i32;
    pub fn mi355_flat_close(flat: *mut mi355_flat) -> i32;
    pub fn mi355_flat_set_stream(flat: *mut mi355_flat, hip_stream: *mut c_void) -> i32;
    pub fn mi355_flat_sync(flat: *mut mi355_flat) -> i32;
    pub fn mi355_flat_search(flat: *mut mi355_flat, queries: *const f32, n_queries: u32, params: *const mi355_search_params, out_rowids: *mut u64, out_dist: *mut f32, out_counts: *mut u32) -> i32;
    pub fn mi355_flat_configure(flat: *mut mi355_flat, gemm_variant: u32, grid_workgroups: u32, flags: u32) -> i32;
    pub fn mi355_flat_checksum(flat: *mut mi355_flat, out_checksum: *mut u64) -> i32;
    pub fn mi355_flat_census(flat: *mut mi355_flat, out_never_filter: *mut u64, out_not_finite: *mut u64, out_sum: *mut f64) -> i32;
    pub fn mi355_flat_last_stats(flat: *mut mi355_flat, out: *mut mi355_flat_stats) -> i32;
    pub fn mi355_flat_info(flat: *const mi355_flat, out_last_path: *mut u32, out_has_filter: *mut u32) -> i32;
    pub fn mi355_merge_topk(device: i32, hip_stream: *mut c_void, in_rowids: *const u64, in_dist: *const f32, in_counts: *const u32, n_lists: u32, n_queries: u32, k: u32, out_rowids: *mut u64, out_dist: *mut f32, out_counts: *mut u32) -> i32;
    pub fn mi355_ivfpq_encode(desc: *const mi355_encode_desc, vectors: *const f32, n_rows: u64, out_part_offsets: *mut u64, out_codes: *mut u8, out_order: *mut u64, out_assign: *mut u32) -> i32;
    pub fn mi355_kmeans_train(desc: *const mi355_kmeans_desc, vectors: *const f32, n_rows: u64, centroids: *mut f32, out_counts: *mut u64) -> i32;
    pub fn mi355_pq_train(desc: *const mi355_pq_train_desc, residuals: *const f32, n_rows: u64, codebook: *mut f32) -> i32;
    pub fn mi355_ivf_residuals(desc: *const mi355_kmeans_desc, vectors: *const f32, n_rows: u64, centroids: *const f32, out_residuals: *mut f32, out_assign: *mut u32) -> i32;
    pub fn mi355_comm_unique_id(out_id: *mut c_void) -> i32;
    pub fn mi355_comm_create(id: *const c_void, rank: u32, world: u32, device: i32, out: *mut *mut mi355_comm) -> i32;
    pub fn mi355_comm_create_loopback(world: u32, device: i32, out: *mut *mut mi355_comm) -> i32;
    pub fn mi355_comm_destroy(comm: *mut mi355_comm) -> i32;
    pub fn mi355_comm_last_stats(comm: *mut mi355_comm, out: *mut mi355_comm_stats) -> i32;
    pub fn mi355_search_sharded(index: *mut mi355_index, comm: *mut mi355_comm, queries: *const f32, n_queries: u32, params: *const mi355_search_params, flags: u32, out_rowids: *mut u64, out_dist: *mut f32, out_counts: *mut u32) -> i32;
    pub fn mi355_flat_search_sharded(flat: *mut mi355_flat, comm: *mut mi355_comm, queries: *const f32, n_queries: u32, params: *const mi355_search_params, out_rowids: *mut u64, out_dist: *mut f32, out_counts: *mut u32) -> i32;
    pub fn mi355_coarse_slice(nlist: u32, world: u32, rank: u32, out_lo: *mut u32, out_hi: *mut u32) -> i32;
    pub fn mi355_shard_plan(part_offsets: *const u64, nlist: u32, shard_count: u32, out_owner: *mut u32) -> i32;
    pub fn mi355_shard_plan_weighted(part_offsets: *const u64, weight: *const f32, nlist: u32, shard_count: u32, out_owner: *mut u32) -> i32;
}

/// `lancedb::Error` of a non-zero status (error.rs:55-145), with the library's message for this thread.
pub fn status_to_error(status: i32) -> Option<(i32, String)> {
    if status == 0 {
        return None;
    }
    let mut buf = [0 as c_char; 512];
    unsafe { mi355_last_error(buf.as_mut_ptr(), buf.len()) };
    let msg = unsafe { std::ffi::CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned();
    Some((status, msg))
}
