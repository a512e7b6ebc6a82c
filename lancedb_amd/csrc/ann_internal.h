// ann_internal.h — host-side internals shared by the translation units of
// libmi355_ann.so (include/mi355_ann.h is the only public surface).
//
//   ann_core.hip          library calls, error slot, shard plan
//   ann_index_open.hip    IVF-PQ handle lifecycle: open (packing, planner tables) / close / configure / raw column
//   ann_index.hip         the search pipeline of one device batch (run_ivfpq), stage timers, statistics
//   ann_index_search.hip  call driver: request checks, coalescing queue, graph cache, host I/O, mi355_search*
//   ann_scan_skew*.hip    launchers / instantiations of the production scan kernel (plain and padded / multi-slab)
//   ann_scan_pair.hip     launcher of the generic scan kernel (4-bit codes, MI355_INDEX_GENERIC_SCAN)
//   ann_flat.hip          flat handle: open / search (MFMA filter + exact re-rank)
//   ann_build.hip   index training and population
//   ann_comm.hip    RCCL exchange behind the ABI: mi355_comm_*, mi355_search_sharded
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "../../include/mi355_ann.h"
#include "call_queue.h"
#include "device_common.h"

// ------------------------------------------------------------------ errors --
// message of the last failing call on this thread (mi355_last_error)
int32_t fail(int32_t code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Exception barrier of the C ABI (include/mi355_ann.h: "no exception or abort crosses the ABI"; the statuses are
// rust/lancedb/src/error.rs:55-145's Runtime).  Every extern "C" entry point is a function-try-block closed by this
// macro: a std::bad_alloc from a host container, any other C++ exception, or a foreign one becomes status 2 with a
// message in the per-thread slot (which does not allocate) instead of unwinding into a Rust / C caller (undefined
// behaviour there).  tests/test_abi.py::test_out_of_host_memory_is_a_status_not_a_crash runs an entry point under RLIMIT_AS.
#define MI355_ABI_GUARD(name)                                                                            \
  catch (const std::bad_alloc&) { return fail(MI355_ERR_RUNTIME, name ": out of host memory"); }          \
  catch (const std::exception& e_) { return fail(MI355_ERR_RUNTIME, name ": C++ exception: %s", e_.what()); } \
  catch (...) { return fail(MI355_ERR_RUNTIME, name ": unknown C++ exception"); }

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess)                                                               \
      return fail(MI355_ERR_RUNTIME, "HIP error %d (%s) at %s:%d: %s", (int)_e,         \
                  hipGetErrorString(_e), __FILE__, __LINE__, #expr);                    \
  } while (0)

#define ST_TRY(expr)               \
  do {                             \
    int32_t _s = (expr);           \
    if (_s != MI355_OK) return _s; \
  } while (0)

// grow-only device buffer.  Handle members are released by the handle's close; a
// function-local one is a ScratchBuf (released on every exit path).
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  uint32_t* gen = nullptr;  // bumped on every re-allocation (invalidates cached hipGraphs)
  int32_t ensure(size_t bytes) {
    if (bytes <= cap) return MI355_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      p = nullptr;
      (void)hipGetLastError();
      return fail(MI355_ERR_RUNTIME, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    cap = want;
    if (gen) ++*gen;
    return MI355_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T* as() const {
    return (T*)p;
  }
};

struct ScratchBuf : DevBuf {
  ScratchBuf() = default;
  ScratchBuf(const ScratchBuf&) = delete;
  ScratchBuf& operator=(const ScratchBuf&) = delete;
  ~ScratchBuf() { release(); }
};

static inline size_t dtype_size(uint32_t dt) { return dt == MI355_DTYPE_F32 ? 4 : 2; }

// copy `bytes` from a caller buffer (host or device) to device memory
static inline hipError_t copy_in(void* dst, const void* src, size_t bytes, uint32_t mem, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  return hipMemcpyAsync(dst, src, bytes, mem == MI355_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s);
}

int32_t need_device(int32_t device);
void shard_plan_host(const uint64_t* po, uint32_t nlist, uint32_t shards, std::vector<uint32_t>& owner,
                     const float* weight = nullptr);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize), once per kernel, device and size.  The call costs the host tens of microseconds;
// made on every search it pushed the submission of the launches behind it past the END of a short scan (round 6: behind a 43 us scan
// at the reference's default index shape the merge kernel "ran" 59 us — waiting to be submitted).
static inline hipError_t ensure_dyn_lds(const void* kern, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> granted;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> g(mu);
  size_t& have = granted[{kern, dev}];
  if (have >= bytes) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) have = bytes;
  return e;
}

// Build-time dev knobs (-DMI355_DEV_KNOBS, scripts/build_variants.sh only): environment
// overrides for kernel tuning experiments.  The product build reads no environment.
#ifdef MI355_DEV_KNOBS
static inline uint32_t dev_knob(const char* name, uint32_t dflt) {
  const char* s = getenv(name);
  if (!s || !*s) return dflt;
  return (uint32_t)strtoul(s, nullptr, 10);
}
#else
static inline uint32_t dev_knob(const char*, uint32_t dflt) { return dflt; }
#endif

// CUs the scans of a handle leave free while its deferred re-rank (a PCIe gather) runs beside them
#ifndef MI355_REFINE_SIDE_CUS
#define MI355_REFINE_SIDE_CUS 16u
#endif
// ... and the four-wave workgroups of that re-rank per reserved CU (no LDS, ~80 VGPRs: up to six fit)
#ifndef MI355_REFINE_SIDE_WGS_PER_CU
#define MI355_REFINE_SIDE_WGS_PER_CU 1u
#endif

// top-k selection width: slots per lane of the in-register selector (64 lanes each);
// k > 256 runs the same selector in several passes (device_common.h, WaveTopK floor)
static inline int kpl_for(uint32_t kk) { return kk <= 64 ? 1 : kk <= 128 ? 2 : 4; }

template <typename Args, typename KernFn>
static inline void launch_by_kpl(int kpl, KernFn k1, KernFn k2, KernFn k4, dim3 grid, dim3 block, size_t lds,
                                 hipStream_t st, const Args& a) {
  if (kpl == 1)
    hipLaunchKernelGGL(k1, grid, block, lds, st, a);
  else if (kpl == 2)
    hipLaunchKernelGGL(k2, grid, block, lds, st, a);
  else
    hipLaunchKernelGGL(k4, grid, block, lds, st, a);
}

// ---------------------------------------------------------------- handles ---
// Per-launch-sequence timestamps, recorded on the search stream without any
// host synchronisation; elapsed times are read back in mi355_last_stats.
struct EventSet {
  hipEvent_t ev[7];  // [6]: between the planner (+ the batch's distance-table images) and the scan kernel
};


struct GraphKey {
  uint32_t nq, k, kk, nprobe, flags;  // flags: refine | range shape | filter mode
  bool operator<(const GraphKey& o) const {
    return std::tie(nq, k, kk, nprobe, flags) < std::tie(o.nq, o.k, o.kk, o.nprobe, o.flags);
  }
};
struct GraphEntry {
  hipGraphExec_t exec = nullptr;
  bool seen = false;    // the shape ran eagerly once (the workspace is sized): the next call captures
  bool failed = false;  // capture / instantiate did not work for this shape: eager from now on
  uint32_t gen = 0;     // workspace generation the graph was captured against
  // what is baked into the captured kernels (a replay needs them to be identical)
  float lower = 0.f, upper = 0.f;  // (the deadline is NOT baked in: replays arm the control word eagerly)
  const void* d_q = nullptr;
  const void* d_ids = nullptr;
  uint64_t work_items = 0;
};

// one waiting host-I/O search of the coalescing queue (call_queue.h)
struct PendingSearch : QueueWaiter {
  const float* queries;
  const mi355_search_params* params;
  uint64_t* out_rowids;
  float* out_dist;
  uint32_t* out_counts;
  std::chrono::steady_clock::time_point t0;  // when the call entered the library (its timeout starts there)
};

struct mi355_index {
  int32_t device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::mutex mu;  // serialises device work of the handle
  // shape
  uint32_t dim = 0, nlist = 0, m = 0, dsub = 0, metric = 0, nbits = 8, mb = 0;  // mb: code bytes per row
  uint64_t n_local = 0;
  uint32_t parts_owned = 0, max_len = 0;
  uint32_t shard_count = 1, shard_rank = 0;
  // device data
  DevBuf centroids, cnorm, codebook, codes, code_off, plen, pstride, lrow0, grow0, row_ids, raw;
  bool has_row_ids = false, has_raw = false, local_arrays = false;
  uint32_t raw_dtype = 0;
  const void* raw_attached = nullptr;  // mi355_index_attach_raw: borrowed device column (local row order)
  uint32_t raw_attached_dtype = 0;
  void* raw_mapped_host = nullptr;  // MI355_INDEX_RAW_HOST_MAPPED: registered caller memory
  const void* raw_mapped_dev = nullptr;
  std::vector<uint64_t> raw_row_of_local;  // unused unless mapped (see ann_index.hip)
  std::vector<uint32_t> h_plen;
  // code layout: MI355_SCAN_PAIR = [mb][pstride] blocks, MI355_SCAN_SKEW = pre-skewed streams
  uint32_t layout = MI355_SCAN_PAIR;
  uint32_t n_cus = 256;
  bool merge_block_tried = false, merge_block_ok = false;  // k_merge_cands<KPL, 16> may use MERGE_BLOCK_LDS bytes
  uint32_t wall_khz = 100000;  // rate of the constant device clock behind timeout_ms (wall_clock64)
  DevBuf cbT, order, xcd_first, p_cnt, p_off, p_fill, q_start, heads, items, qthr, w_filter, w_probes64;
  // SkewShape of the packed codes (MI355_SCAN_SKEW): columns per slab, slabs per row, generalised kernel or not
  uint32_t sk_M = 0, sk_slabs = 1, sk_slabbed = 0, sk_res_floats = 0;
  DevBuf w_partial;  // per-workgroup partial row sums between the slabs of a work item (sk_slabs > 1)
  // batch-level distance tables (kernels_lut.h): the shape qualifies (8-bit codes, sub-vectors of 16 floats: the
  // reference's m = dim / 16), the per-pair residual rows and the table images of a chunk of the batch
  bool lut_img_ok = false;
  bool lut_inline_cfg = false;  // MI355_CFG_LUT_INLINE: build the tables inside the scan work items anyway
  DevBuf w_lutres, w_lutimg;
  DevBuf w_tl;           // (-DMI355_DEV_TIMELINE builds) the last scan launch's per-workgroup stamps
  uint32_t tl_grid = 0;
  DevBuf w_cand2b, w_cnt2b;  // the second buffer set of the deferred refine (each set has its own allocations)
  bool defer_cfg = false;    // MI355_CFG_DEFER_REFINE
  // workspace
  DevBuf w_q, w_qp, w_qq, w_coarse, w_probes, w_cand, w_ids, w_dist, w_pos, w_cnt, w_ids2, w_dist2, w_cnt2, w_ctl,
      w_cand2, w_sq, w_sids, w_sdist, w_scnt, w_scnt_ann, w_spill, w_srows, w_ccnt;
  uint32_t ws_gen = 0;  // bumped by every workspace re-allocation
  uint32_t second_np = 0;  // nprobe of the last maximum_nprobes second pass (stats: DevCtl.short_queries x this)
  // overlapped sharded search (ann_comm.hip): the exchange of the last call may still run on the
  // communicator's stream; `xdone` is recorded behind it and every other entry point joins it first
  hipEvent_t xdone = nullptr;
  bool xpending = false;
  void* h_pin = nullptr;  // page-locked staging block of small host-I/O batches (queries in, results out)
  size_t h_pin_cap = 0;
  // deferred refine (device-I/O calls): the exact re-rank of call i runs on `rstream` under the scan of call i+1
  // (a host-mapped raw column makes it PCIe-bound while the scan is LDS / VALU-bound); two sets of its buffers
  bool raw_is_host = false;  // the refine column is host memory (mapped): only then is the re-rank deferred
  hipStream_t rstream = nullptr;
  hipEvent_t r_scan[2] = {nullptr, nullptr}, r_done[2] = {nullptr, nullptr};
  bool r_busy[2] = {false, false};
  uint64_t r_seq = 0;
  // config
  uint32_t scan_variant = MI355_SCAN_AUTO, slice_rows = 0, profile = 0;
  bool use_graph = false;  // graph replay measured slower than eager launches
  std::atomic<bool> coalesce{true};  // read without the handle's lock by every host-I/O call
  mi355_stats stats{};
  std::vector<EventSet> ev_free, ev_pending;
  std::map<GraphKey, GraphEntry> graphs;
  CallQueue<PendingSearch> cq;  // coalescing queue of concurrent host-I/O callers (call_queue.h)
};

struct mi355_flat {
  int32_t device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::mutex mu;
  uint32_t dim = 0, dtype = 0;
  uint64_t n_rows = 0;
  DevBuf vectors, row_ids;
  bool has_row_ids = false;
  DevBuf w_q, w_cand, w_ids, w_dist, w_cnt;
  void* h_pin = nullptr;  // page-locked staging of small host-I/O calls (queries in, the three result arrays out)
  size_t h_pin_cap = 0;
  // MFMA filter + exact re-rank (kernels_flat_mfma.h)
  bool mfma = false;      // built at open when the column is large enough
  bool shadowed = false;  // GEMM reads a bf16 shadow (column is f32/f16 or dim % 64 != 0)
  uint32_t dimp = 0;
  float c_err = 0.f, vv_max = 0.f;
  DevBuf shadow, vv, vw_cos, vw_dot, vmax, g_qb, g_qa, g_qg, g_slack, g_tau, g_gm, g_seg, g_cnt, g_cand, w_filter, w_sum, w_fallback;
  uint32_t last_path = 0;  // 1 = MFMA filter, 2 = exact sweep (reported by mi355_flat_info)
  uint32_t gemm_variant = MI355_FLAT_GEMM_AUTO, grid_workgroups = 0, cfg_flags = 0;
  uint64_t checksum = 0, census[2] = {0, 0};
  double census_sum = 0.0;
  // MI355_FLAT_PROFILE: events {before prep, before GEMM, after GEMM, after re-rank} per launch sequence
  struct FlatEv {
    hipEvent_t ev[4];
  };
  std::vector<FlatEv> ev_free, ev_pending;
  mi355_flat_stats fstats{};
};

// ------------------------------------------------- cross-TU entry points ----
// one pass of the IVF-PQ pipeline over `nq` device-resident queries (ann_index.hip)
struct SearchPlan {
  uint32_t k, kk, nprobe;
  bool refine;
  RangeFilter range;
  RowFilter filter;
  const uint64_t* ext_probes = nullptr;  // device [nq, nprobe]: skip the coarse stage (mi355_search_probes)
  // sharded search: stop after the ANN merge and leave the kk best (distance, position, rowid)
  // records per query in out_cand [nq, kk] (refine runs after the cross-rank merge)
  Cand* out_cand = nullptr;
  ActiveMask act;      // device-side batch size: the maximum_nprobes second pass (slots past *act.n are skipped)
  uint32_t ws_mb = 0;  // workspace budget of this pass in MiB (0 = the default)
  bool defer_refine = false;  // run refine + final merge on the handle's refine stream (results complete at mi355_index_sync)
  uint32_t rset = 0;          // which of the two refine buffer sets this call uses (deferred calls alternate)
  // the call's control word is armed by the pipeline's first kernel (k_coarse_lat) instead of a k_arm_deadline launch
  bool arm_in_front = false;
  unsigned long long arm_ticks = 0;
  uint32_t arm_reset = 0;
  // a single host query handed to the latency front in its kernel arguments (no staging copy): the HOST pointer
  const float* host_q = nullptr;
};
// the latency front (k_coarse_lat + k_select_plan) applies to this pass: a handful of queries on the production scan
bool lat_front_applies(const mi355_index* ix, uint32_t nq, const SearchPlan& pl);
int32_t run_ivfpq(mi355_index* ix, const float* d_q, uint32_t nq, const SearchPlan& pl, uint64_t* d_ids,
                  float* d_dist, uint32_t* d_cnt, uint32_t* d_cnt_ann);
int32_t validate_params(const mi355_search_params* p);
int32_t make_row_filter(const mi355_search_params* p, DevBuf& stage, hipStream_t st, RowFilter* out);
int32_t arm_deadline(mi355_index* ix, uint32_t timeout_ms, hipStream_t st);

// scan launchers (ann_scan_pair.hip / ann_scan_skew.hip)
struct ScanArgs;
struct SkewArgs;
int32_t launch_scan_pair(const ScanArgs& sa, dim3 grid, hipStream_t st, uint32_t vpt, uint32_t nt);
size_t scan_pair_lds(uint32_t m, uint32_t nbits, uint32_t dim, uint32_t lr, uint32_t nt);
uint32_t scan_pair_m_lds(uint32_t m, uint32_t nbits, uint32_t dim);
int32_t launch_scan_skew(const SkewArgs& sa, uint32_t M, uint32_t slabbed, uint32_t n_blocks, uint64_t n_items, uint32_t kk, hipStream_t st);
// ... the kernels that copy a table image (SkewArgs::lut_img) instead of building the table (ann_scan_skew_img.hip), and the
// batch-level table kernels themselves (ann_lut.hip)
int32_t launch_scan_skew_img(const SkewArgs& sa, uint32_t M, uint32_t slabbed, uint32_t n_blocks, uint64_t n_items, uint32_t kk, hipStream_t st);
int32_t launch_scan_skew_lat(const SkewArgs& sa, uint32_t M, uint32_t n_blocks, uint64_t n_items, uint32_t kk, hipStream_t st);
struct SkewItem;
bool lut_images_shape_ok(const mi355_index* ix);
size_t lut_image_bytes_per_pair(const mi355_index* ix);
size_t lut_residual_bytes_per_pair(const mi355_index* ix);
int32_t launch_lut_images(mi355_index* ix, const float* qp, const SkewItem* items, const uint32_t* q_start, uint32_t n_pairs,
                          uint32_t n_slices, uint32_t nprobe, float* res, float* img, hipStream_t st);

struct IndexView;
int32_t launch_refine(mi355_index* ix, const IndexView& view, const float* q, uint32_t nq, const Cand* in,
                      const uint32_t* in_cnt, const uint32_t* owner, uint32_t my_rank, uint32_t kk,
                      const RangeFilter& range, Cand* out, hipStream_t st, ActiveMask act = ActiveMask(),
                      uint32_t max_blocks_y = 0);
IndexView make_view(const mi355_index* ix);

// pieces of the search path shared with the sharded search (ann_comm.hip)
struct SearchShape {
  uint32_t k, kk, np_min, np_max;
};
int32_t check_search(mi355_index* ix, const float* queries, uint32_t n_queries, const mi355_search_params* p,
                     const uint64_t* ext_probes, uint32_t ext_nprobe, uint64_t* out_rowids, float* out_dist,
                     uint32_t* out_counts, SearchShape* sh, bool sharded_call);
void account(mi355_index* ix, uint32_t nq, uint32_t nprobe);
int32_t drain_events(mi355_index* ix, bool discard);
int32_t join_exchange(mi355_index* ix);  // make ix->stream wait for an exchange still running on a communicator's stream
int32_t expand_short_device(mi355_index* ix, const uint32_t* d_cnt_ann, uint32_t n_queries, uint32_t kk, const float* d_q,
                            DevBuf& rows, DevBuf& sq, hipStream_t st, ActiveMask* out_act);
void reset_stats(mi355_index* ix);
int32_t coarse_topn_device(mi355_index* ix, const float* d_q, uint32_t nq, uint32_t nprobe, uint32_t cent_lo,
                           uint32_t cent_hi, uint64_t* d_ids, float* d_dist, uint32_t* d_cnt);
int32_t run_flat_search_device(mi355_flat* f, const float* d_q, uint32_t nq, const mi355_search_params* p,
                               uint64_t* d_ids, float* d_dist, uint32_t* d_cnt);
