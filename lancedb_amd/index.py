"""Device-resident index handles over the C ABI (include/mi355_ann.h).

`IvfPqIndex` plays the role the lance `Session` index cache plays in the
reference (python/src/session.rs:49-50): open once, search many times.
Arrays may be numpy (host) or anything exposing a device pointer through
`data_ptr()` (e.g. a torch tensor on the GPU) — only raw pointers cross the ABI.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._lib import check, lib


def _is_device(a):
    return hasattr(a, "data_ptr") and getattr(a, "is_cuda", False)


def _ptr(a):
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    return a.ctypes.data_as(C.c_void_p)


def _device_empty_like(ref, specs):
    """Uninitialised device arrays of the same kind as `ref` (torch tensor or
    lancedb_amd.DeviceArray)."""
    from ._hip import DeviceArray
    if isinstance(ref, DeviceArray):
        return [DeviceArray(shape, dt, ref.device) for shape, dt in specs]
    import torch
    return [torch.empty(shape, dtype=getattr(torch, dt), device=ref.device) for shape, dt in specs]


def _host(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


_KEEP = object()  # "argument not given": FlatIndex.configure keeps the handle's current value

class SearchResult:
    """Per query: `counts[q]` rows sorted by (_distance, _rowid); padded with
    UINT64_MAX / +inf (python/python/lancedb/query.py:1365-1370)."""

    def __init__(self, rowids, distances, counts):
        self.rowids, self.distances, self.counts = rowids, distances, counts

    def __iter__(self):
        return iter((self.rowids, self.distances, self.counts))


class _Handle:
    _close_fn = None

    def __init__(self):
        self._h = C.c_void_p()
        self._keep = []

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            getattr(lib(), self._close_fn)(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _run_search(fn, handle, dim, queries, params, out=None):
    """Shared host/device marshalling of mi355_search / mi355_flat_search."""
    if _is_device(queries):
        q = queries.contiguous().view(-1, dim)
        nq, k = q.shape[0], params.k
        if out is None:
            ids, dist, cnt = _device_empty_like(q, [((nq, k), "int64"), ((nq, k), "float32"), ((nq,), "int32")])
        else:
            ids, dist, cnt = out
        params.io_mem = _abi.MEM_DEVICE
        check(fn(handle, _ptr(q), C.c_uint32(nq), C.byref(params), _ptr(ids), _ptr(dist), _ptr(cnt)))
        return SearchResult(ids, dist, cnt)
    q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, dim)
    nq, k = q.shape[0], params.k
    ids = np.empty((nq, k), dtype=np.uint64)
    dist = np.empty((nq, k), dtype=np.float32)
    cnt = np.zeros(nq, dtype=np.uint32)
    params.io_mem = _abi.MEM_HOST
    check(fn(handle, _ptr(q), C.c_uint32(nq), C.byref(params), _ptr(ids), _ptr(dist), _ptr(cnt)))
    return SearchResult(ids, dist, cnt)


class IvfPqIndex(_Handle):
    """An IVF-PQ index resident on one MI355X.

    Parameters mirror IvfPqIndexBuilder's outputs
    (rust/lancedb/src/index/vector.rs:266-319): `centroids` [nlist, dim] f32,
    `codebook` [m, 256, dim/m] f32, `part_offsets` [nlist+1], `codes` u8 in
    `codes_layout`, optional `row_ids` / `raw_vectors` in index order.
    """
    _close_fn = "mi355_index_close"

    def __init__(self, centroids, codebook, part_offsets, codes, row_ids=None, raw_vectors=None,
                 metric="l2", codes_layout=_abi.CODES_ROW_MAJOR, raw_dtype=_abi.DTYPE_F32,
                 device=0, shard_count=1, shard_rank=0, nbits=8, generic_scan=False, raw_host_mapped=False,
                 local_arrays=False, part_owner=None):
        """nbits: 8, or 4 (codebook [m, 16, dim/m], codes [n, m/2] with sub-quantiser 2t in the low
        nibble of byte t; table/create_index.rs:96-101).  generic_scan: keep the generic code layout
        (k_scan_pair) for an m the production scan supports.  raw_host_mapped: `raw_vectors` (a host
        array the caller keeps alive) is page-locked and read zero-copy by the refine stage.
        local_arrays: `codes` / `row_ids` / `raw_vectors` hold only the partitions this shard owns
        (`shard_plan(part_offsets, shard_count) == shard_rank`), concatenated in partition order."""
        super().__init__()
        on_dev = _is_device(codes)
        po = np.ascontiguousarray(part_offsets, dtype=np.uint64)  # always host
        if on_dev:
            cen, cb, cd, rid, raw = centroids, codebook, codes, row_ids, raw_vectors
            nlist, dim = cen.shape
            m = cb.shape[0]
        else:
            cen = _host(centroids, np.float32)
            cb = _host(codebook, np.float32)
            cd = _host(codes, np.uint8)
            rid = _host(row_ids, np.uint64)
            raw = None if raw_vectors is None else np.ascontiguousarray(raw_vectors)
            nlist, dim = cen.shape
            m = cb.shape[0]
        self.dim, self.nlist, self.m = int(dim), int(nlist), int(m)
        self.metric = _abi.METRIC_NAMES[metric] if isinstance(metric, str) else int(metric)
        d = _abi.IndexDesc()
        d.struct_size = C.sizeof(_abi.IndexDesc)
        d.dim, d.nlist, d.m, d.nbits = self.dim, self.nlist, self.m, int(nbits)
        d.flags = ((_abi.INDEX_GENERIC_SCAN if generic_scan else 0) | (_abi.INDEX_RAW_HOST_MAPPED if raw_host_mapped else 0)
                   | (_abi.INDEX_LOCAL_ARRAYS if local_arrays else 0))
        self.nbits = int(nbits)
        d.metric = self.metric
        d.n_rows = int(po[-1])
        d.mem = _abi.MEM_DEVICE if on_dev else _abi.MEM_HOST
        d.codes_layout = codes_layout
        d.centroids, d.codebook, d.part_offsets = _ptr(cen), _ptr(cb), _ptr(po)
        d.codes, d.row_ids, d.raw_vectors = _ptr(cd), _ptr(rid), _ptr(raw)
        d.raw_dtype = raw_dtype
        d.device = device
        d.shard_count, d.shard_rank = shard_count, shard_rank
        if part_owner is not None:  # [nlist] shard of every partition (shard_plan(..., weights=...)); host memory
            self._owner = np.ascontiguousarray(part_owner, dtype=np.uint32)
            if self._owner.shape != (int(d.nlist),):
                raise ValueError("part_owner must be [nlist]")
            d.part_owner = self._owner.ctypes.data_as(C.c_void_p)
        self.n_rows = d.n_rows
        self._keep = [cen, cb, po, cd, rid, raw]
        check(lib().mi355_index_open(C.byref(d), C.byref(self._h)))
        self._keep = [raw] if raw_host_mapped else []  # the library copied everything else it needs

    def configure(self, scan_variant=None, slice_rows=None, profile=0, graph=None, coalesce=None, defer_refine=None, lut_inline=None):
        """profile: 0 counters only, 1 per-stage times of the last search,
        2 accumulate over searches until the next configure().  graph / coalesce: the
        hipGraph-replay and coalescing-queue modes of host-I/O searches; None keeps the current
        setting (after open: coalescing on, graph replay off — measured slower than eager launches
        on the driver's box, profiles/r02 latency leg).  defer_refine (opt-in, MI355_CFG_DEFER_REFINE): device-I/O refine
        calls over a host-resident raw column leave their re-rank on a private stream — results complete at sync().
        lut_inline (opt-out, MI355_CFG_LUT_INLINE): build every PQ distance table inside its scan work item even where the
        batch-level table kernel applies (dim / m = 16); results are bit-identical either way."""
        # None keeps the handle's current value (a profile-only call such as analyze_plan() must not reset the tuning)
        if scan_variant is not None:
            self._scan_variant = int(scan_variant)
        if slice_rows is not None:
            self._slice_rows = int(slice_rows)
        scan_variant = getattr(self, "_scan_variant", _abi.SCAN_AUTO)
        slice_rows = getattr(self, "_slice_rows", 0)
        if defer_refine is not None:
            self._defer = bool(defer_refine)
        if lut_inline is not None:
            self._lut_inline = bool(lut_inline)
        if graph is not None:
            self._graph = bool(graph)
        if coalesce is not None:
            self._coalesce = bool(coalesce)
        mode = int(profile) | (_abi.CFG_GRAPH if getattr(self, "_graph", False) else 0) | \
            (_abi.CFG_COALESCE if getattr(self, "_coalesce", True) else 0) | (_abi.CFG_DEFER_REFINE if getattr(self, "_defer", False) else 0) | \
            (_abi.CFG_LUT_INLINE if getattr(self, "_lut_inline", False) else 0)
        check(lib().mi355_index_configure(self._h, C.c_uint32(scan_variant), C.c_uint32(slice_rows), C.c_uint32(mode)))

    def set_stream(self, hip_stream):
        check(lib().mi355_index_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def sync(self):
        check(lib().mi355_index_sync(self._h))

    def attach_raw_vectors(self, raw, raw_dtype=_abi.DTYPE_F32):
        """Borrow a DEVICE array [rows on this handle, dim] (local row order) as the refine column,
        without copying it; the caller keeps it alive until detach_raw_vectors() / close()."""
        if not _is_device(raw):
            raise ValueError("attach_raw_vectors takes a device array (open with raw_host_mapped for host columns)")
        check(lib().mi355_index_attach_raw(self._h, _ptr(raw), C.c_uint32(raw_dtype)))
        self._attached = raw

    def detach_raw_vectors(self):
        check(lib().mi355_index_detach_raw(self._h))
        self._attached = None

    def info(self):
        rows, parts = C.c_uint64(0), C.c_uint32(0)
        check(lib().mi355_index_info(self._h, C.byref(rows), C.byref(parts)))
        return rows.value, parts.value

    def stats(self):
        s = _abi.Stats()
        s.struct_size = C.sizeof(_abi.Stats)
        check(lib().mi355_last_stats(self._h, C.byref(s)))
        return {name: getattr(s, name) for name, _ in _abi.Stats._fields_ if name != "struct_size"}

    def search(self, queries, params=None, out=None, **kw):
        """queries [nq, dim] f32 (numpy or device tensor) -> SearchResult."""
        p = params if params is not None else _abi.make_params(**kw)
        return _run_search(lib().mi355_search, self._h, self.dim, queries, p, out)

    # ---- two-phase search (sharded coarse stage) ----------
    def coarse_topn(self, queries, nprobe, cent_lo=0, cent_hi=None):
        """Best min(nprobe, slice) partitions of centroid slice [cent_lo, cent_hi) per query
        -> (part_ids [nq, nprobe] u64 / int64, dist [nq, nprobe] f32, counts [nq]) in
        merge_topk's list layout (padding: UINT64_MAX / +inf)."""
        cent_hi = self.nlist if cent_hi is None else cent_hi
        if _is_device(queries):
            q = queries.contiguous().view(-1, self.dim)
            nq = q.shape[0]
            ids, dist, cnt = _device_empty_like(q, [((nq, nprobe), "int64"), ((nq, nprobe), "float32"), ((nq,), "int32")])
            mem = _abi.MEM_DEVICE
        else:
            q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.dim)
            nq = q.shape[0]
            ids = np.empty((nq, nprobe), dtype=np.uint64)
            dist = np.empty((nq, nprobe), dtype=np.float32)
            cnt = np.zeros(nq, dtype=np.uint32)
            mem = _abi.MEM_HOST
        check(lib().mi355_coarse_topn(self._h, _ptr(q), C.c_uint32(nq), C.c_uint32(nprobe), C.c_uint32(cent_lo),
                                      C.c_uint32(cent_hi), C.c_uint32(mem), _ptr(ids), _ptr(dist), _ptr(cnt)))
        return ids, dist, cnt

    def search_probes(self, queries, probes, params=None, out=None, **kw):
        """Scan the partitions of `probes` [nq, nprobe] (u64 ids, e.g. merge_topk of the
        gathered coarse_topn lists) that this handle owns."""
        p = params if params is not None else _abi.make_params(**kw)
        nprobe = int(probes.shape[1])
        if not _is_device(queries):
            probes = np.ascontiguousarray(probes, dtype=np.uint64)

        def fn(handle, q, nq, params_ref, ids, dist, cnt):
            return lib().mi355_search_probes(handle, q, nq, params_ref, _ptr(probes), C.c_uint32(nprobe), ids, dist, cnt)
        return _run_search(fn, self._h, self.dim, queries, p, out)


class FlatIndex(_Handle):
    """A raw vector column on the GPU for exhaustive search
    (bypass_vector_index, rust/lancedb/src/query.rs:1360-1370)."""
    _close_fn = "mi355_flat_close"

    def __init__(self, vectors, row_ids=None, dtype=_abi.DTYPE_F32, device=0):
        super().__init__()
        on_dev = _is_device(vectors)
        if on_dev:
            v, rid = vectors, row_ids
        else:
            v = np.ascontiguousarray(vectors)
            if dtype == _abi.DTYPE_F32:
                v = np.ascontiguousarray(v, dtype=np.float32)
            rid = _host(row_ids, np.uint64)
        n, dim = v.shape
        self.dim, self.n_rows = int(dim), int(n)
        d = _abi.FlatDesc()
        d.struct_size = C.sizeof(_abi.FlatDesc)
        d.dim, d.n_rows, d.dtype = self.dim, self.n_rows, dtype
        d.mem = _abi.MEM_DEVICE if on_dev else _abi.MEM_HOST
        d.vectors, d.row_ids, d.device = _ptr(v), _ptr(rid), device
        check(lib().mi355_flat_open(C.byref(d), C.byref(self._h)))

    def set_stream(self, hip_stream):
        check(lib().mi355_flat_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def sync(self):
        check(lib().mi355_flat_sync(self._h))

    def configure(self, gemm_variant=_KEEP, grid_workgroups=_KEEP, checksum=_KEEP, profile=False, path=_KEEP):
        """Tuning of the GEMM filter (include/mi355_ann.h mi355_flat_configure).  `path`: None = the cheaper exact path
        per call, "filter" = MFMA filter + exact re-rank whenever the call allows it, "sweep" = the exact sweep.
        An argument that is NOT given keeps the handle's current value (after open: automatic schedule / grid / path, no
        checksum), so a profile-only call — analyze_plan() — does not undo a pinned A/B configuration;
        `profile` is per call like IvfPqIndex.configure's."""
        if gemm_variant is not _KEEP:
            self._gemm_variant = int(gemm_variant)
        if grid_workgroups is not _KEEP:
            self._grid = int(grid_workgroups)
        if checksum is not _KEEP:
            self._checksum = bool(checksum)
        if path is not _KEEP:
            if path not in (None, "filter", "sweep"):
                raise ValueError("path must be None, 'filter' or 'sweep'")
            self._path = path
        path = getattr(self, "_path", None)
        flags = (_abi.FLAT_CHECKSUM if getattr(self, "_checksum", False) else 0) | (_abi.FLAT_PROFILE if profile else 0)
        flags |= _abi.FLAT_FORCE_FILTER if path == "filter" else _abi.FLAT_FORCE_SWEEP if path == "sweep" else 0
        check(lib().mi355_flat_configure(self._h, C.c_uint32(getattr(self, "_gemm_variant", _abi.FLAT_GEMM_AUTO)),
                                         C.c_uint32(getattr(self, "_grid", 0)), C.c_uint32(flags)))

    def stats(self):
        """GEMM kernel time accumulated since configure(profile=True) (mi355_flat_last_stats)."""
        s = _abi.FlatStats()
        s.struct_size = C.sizeof(_abi.FlatStats)
        check(lib().mi355_flat_last_stats(self._h, C.byref(s)))
        return {name: getattr(s, name) for name, _ in _abi.FlatStats._fields_ if name != "struct_size"}

    def checksum(self):
        v = C.c_uint64(0)
        check(lib().mi355_flat_checksum(self._h, C.byref(v)))
        return v.value

    def census(self):
        """-> (never_filter, not_finite, finite_sum) of the last checksummed search's group-minimum matrix."""
        a, b, s = C.c_uint64(0), C.c_uint64(0), C.c_double(0.0)
        check(lib().mi355_flat_census(self._h, C.byref(a), C.byref(b), C.byref(s)))
        return a.value, b.value, s.value

    def info(self):
        """-> (last_path, has_filter): 1 = MFMA filter + exact re-rank, 2 = exact sweep."""
        a, b = C.c_uint32(0), C.c_uint32(0)
        check(lib().mi355_flat_info(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def search(self, queries, params=None, out=None, **kw):
        kw.setdefault("nprobe_min", 1)
        kw.setdefault("nprobe_max", 1)
        p = params if params is not None else _abi.make_params(**kw)
        return _run_search(lib().mi355_flat_search, self._h, self.dim, queries, p, out)


def merge_topk(in_rowids, in_dist, in_counts, k, stream=0):
    """Device-side k-way merge of [n_lists, nq, k] candidate lists (device
    arrays: torch tensors or DeviceArray): the reducer after the multi-GPU
    all-gather."""
    n_lists, nq = in_counts.shape
    ids, dist, cnt = _device_empty_like(in_dist, [((nq, k), "int64"), ((nq, k), "float32"), ((nq,), "int32")])
    dev_index = getattr(in_dist.device, "index", in_dist.device) or 0
    check(lib().mi355_merge_topk(C.c_int32(dev_index), C.c_void_p(stream), _ptr(in_rowids),
                                 _ptr(in_dist), _ptr(in_counts), C.c_uint32(n_lists),
                                 C.c_uint32(nq), C.c_uint32(k), _ptr(ids), _ptr(dist), _ptr(cnt)))
    return ids, dist, cnt


def shard_plan(part_offsets, shard_count, weights=None):
    """Partition -> shard assignment (greedy longest-first).  `weights` [nlist]: how often each partition is
    probed (e.g. a histogram of `coarse_topn` over a sample of the query load) — the shards are then balanced by
    the rows they scan instead of the rows they hold; pass the result as `IvfPqIndex(part_owner=...)` on every rank."""
    po = np.ascontiguousarray(part_offsets, dtype=np.uint64)
    out = np.empty(po.size - 1, dtype=np.uint32)
    if weights is None:
        check(lib().mi355_shard_plan(_ptr(po), C.c_uint32(po.size - 1), C.c_uint32(shard_count), _ptr(out)))
    else:
        w = np.ascontiguousarray(weights, dtype=np.float32)
        if w.shape != (po.size - 1,):
            raise ValueError("weights must be [nlist]")
        check(lib().mi355_shard_plan_weighted(_ptr(po), _ptr(w), C.c_uint32(po.size - 1), C.c_uint32(shard_count), _ptr(out)))
    return out


def ivfpq_encode(vectors, centroids, codebook, metric="l2", device=0, return_assign=False, nbits=8):
    """Index population (include/mi355_ann.h mi355_ivfpq_encode): assign every row
    of `vectors` [n, dim] f32 to its IVF partition, PQ-encode its residual with the
    trained `codebook` [m, 256, dim/m] and lay the rows out partition by partition —
    the O(N) transform stage behind `Table.create_index(Index::IvfPq(..))`
    (rust/lancedb/src/table/create_index.rs:114-151, :283-303).

    Host arrays in -> host arrays out; device arrays (torch CUDA tensors or
    DeviceArray, all three inputs) -> device outputs.  Returns
    (part_offsets [nlist+1] u64 numpy, codes [n, m] u8, order [n] i64/u64
    [, assign [n]]): exactly what IvfPqIndex takes (row_ids = ids[order],
    raw_vectors = vectors[order])."""
    dev_in = _is_device(vectors)
    if dev_in != _is_device(centroids) or dev_in != _is_device(codebook):
        raise ValueError("vectors, centroids and codebook must live in the same memory")
    if not dev_in:
        vectors = _host(vectors, np.float32)
        centroids = _host(centroids, np.float32)
        codebook = _host(codebook, np.float32)
    n, dim = int(vectors.shape[0]), int(vectors.shape[1])
    nlist, m = int(centroids.shape[0]), int(codebook.shape[0])
    if nbits not in (4, 8) or (nbits == 4 and m % 2):
        raise ValueError("num_bits must be 8, or 4 with an even num_sub_vectors")
    if tuple(centroids.shape) != (nlist, dim) or tuple(codebook.shape)[:2] != (m, 1 << nbits) or m == 0 or dim % m \
            or int(codebook.shape[2]) != dim // m:
        raise ValueError("centroids must be [nlist, dim] and codebook [m, 2^nbits, dim/m]")
    mb = m * nbits // 8
    mcode = _abi.METRIC_NAMES[metric] if isinstance(metric, str) else int(metric)
    po = np.zeros(nlist + 1, dtype=np.uint64)
    if dev_in:
        dev_index = getattr(vectors.device, "index", vectors.device) or 0
        codes, order, assign = _device_empty_like(vectors, [((n, mb), "uint8"), ((n,), "int64"), ((n,), "int32")])
    else:
        dev_index = device
        codes = np.empty((n, mb), dtype=np.uint8)
        order = np.empty(n, dtype=np.uint64)
        assign = np.empty(n, dtype=np.uint32)
    desc = _abi.EncodeDesc(struct_size=C.sizeof(_abi.EncodeDesc), dim=dim, nlist=nlist, m=m, nbits=nbits, metric=mcode,
                           mem=_abi.MEM_DEVICE if dev_in else _abi.MEM_HOST, device=dev_index,
                           centroids=_ptr(centroids), codebook=_ptr(codebook))
    check(lib().mi355_ivfpq_encode(C.byref(desc), _ptr(vectors), C.c_uint64(n), _ptr(po), _ptr(codes), _ptr(order),
                                   _ptr(assign) if return_assign else None))
    return (po, codes, order, assign) if return_assign else (po, codes, order)


def _kmeans_desc(dim, k, metric, iters, ld, dev_in, dev_index):
    mcode = _abi.METRIC_NAMES[metric] if isinstance(metric, str) else int(metric)
    return _abi.KmeansDesc(struct_size=C.sizeof(_abi.KmeansDesc), dim=dim, k=k, metric=mcode, iters=iters,
                           mem=_abi.MEM_DEVICE if dev_in else _abi.MEM_HOST, device=dev_index, reserved0=0, ld=ld)


def kmeans_train(vectors, init_centroids, metric="l2", iters=50, cols=None, device=0):
    """Deterministic Lloyd iterations on the GPU (include/mi355_ann.h
    mi355_kmeans_train): the trainer behind IvfBuildParams / PQBuildParams
    (rust/lancedb/src/index/vector.rs:61-119).  `init_centroids` [k, dim] seeds
    it; `cols=(lo, hi)` trains on that column range of `vectors` in place (a PQ
    sub-quantiser on the residual matrix).  -> (centroids [k, dim], counts [k])."""
    dev_in = _is_device(vectors)
    if dev_in != _is_device(init_centroids):
        raise ValueError("vectors and init_centroids must live in the same memory")
    if not dev_in:
        vectors = _host(vectors, np.float32)
        cen = np.array(init_centroids, dtype=np.float32, order="C", copy=True)
    else:
        cen, = _device_empty_like(vectors, [(tuple(init_centroids.shape), "float32")])
        _copy_device(cen, init_centroids)
    n, width = int(vectors.shape[0]), int(vectors.shape[1])
    lo, hi = (0, width) if cols is None else cols
    k, dim = int(cen.shape[0]), int(cen.shape[1])
    if not (0 <= lo < hi <= width) or dim != hi - lo or k == 0:
        raise ValueError("init_centroids must be [k, hi - lo] for the trained column range")
    dev_index = (getattr(vectors.device, "index", vectors.device) or 0) if dev_in else device
    desc = _kmeans_desc(dim, k, metric, iters, width, dev_in, dev_index)
    base = C.c_void_p((vectors.data_ptr() if dev_in else vectors.ctypes.data) + 4 * lo)
    counts = np.zeros(k, dtype=np.uint64)
    if dev_in:
        dcounts, = _device_empty_like(vectors, [((k,), "int64")])
        check(lib().mi355_kmeans_train(C.byref(desc), base, C.c_uint64(n), _ptr(cen), _ptr(dcounts)))
        return cen, dcounts
    check(lib().mi355_kmeans_train(C.byref(desc), base, C.c_uint64(n), _ptr(cen), _ptr(counts)))
    return cen, counts


def pq_train(residuals, init_codebook, metric="l2", iters=50, nbits=8, device=0):
    """All PQ sub-quantisers in one call (include/mi355_ann.h mi355_pq_train): `residuals` [n, dim],
    `init_codebook` [m, 2^nbits, dim/m] -> trained codebook of the same shape.  Equal, bit for bit, to
    m kmeans_train(..., cols=(j*dsub, (j+1)*dsub)) calls."""
    dev_in = _is_device(residuals)
    if dev_in != _is_device(init_codebook):
        raise ValueError("residuals and init_codebook must live in the same memory")
    if not dev_in:
        residuals = _host(residuals, np.float32)
        cb = np.array(init_codebook, dtype=np.float32, order="C", copy=True)
    else:
        cb, = _device_empty_like(residuals, [(tuple(init_codebook.shape), "float32")])
        _copy_device(cb, init_codebook)
    n, dim = int(residuals.shape[0]), int(residuals.shape[1])
    m = int(cb.shape[0])
    if m == 0 or dim % m or tuple(cb.shape) != (m, 1 << nbits, dim // m):
        raise ValueError("init_codebook must be [m, 2^nbits, dim/m]")
    mcode = _abi.METRIC_NAMES[metric] if isinstance(metric, str) else int(metric)
    dev_index = (getattr(residuals.device, "index", residuals.device) or 0) if dev_in else device
    desc = _abi.PqTrainDesc(struct_size=C.sizeof(_abi.PqTrainDesc), dim=dim, m=m, nbits=nbits, metric=mcode, iters=iters,
                            mem=_abi.MEM_DEVICE if dev_in else _abi.MEM_HOST, device=dev_index)
    check(lib().mi355_pq_train(C.byref(desc), _ptr(residuals), C.c_uint64(n), _ptr(cb)))
    return cb


def ivf_residuals(vectors, centroids, metric="l2", device=0):
    """x - centroid[partition(x)] for every row (cosine: x normalised first; dot: x):
    the training set of the PQ codebooks.  -> (residuals [n, dim], assign [n])."""
    dev_in = _is_device(vectors)
    if dev_in != _is_device(centroids):
        raise ValueError("vectors and centroids must live in the same memory")
    if not dev_in:
        vectors, centroids = _host(vectors, np.float32), _host(centroids, np.float32)
    n, dim = int(vectors.shape[0]), int(vectors.shape[1])
    k = int(centroids.shape[0])
    if tuple(centroids.shape) != (k, dim) or k == 0:
        raise ValueError("centroids must be [nlist, dim]")
    if dev_in:
        dev_index = getattr(vectors.device, "index", vectors.device) or 0
        out, assign = _device_empty_like(vectors, [((n, dim), "float32"), ((n,), "int32")])
    else:
        dev_index = device
        out, assign = np.empty((n, dim), dtype=np.float32), np.empty(n, dtype=np.uint32)
    desc = _kmeans_desc(dim, k, metric, 0, 0, dev_in, dev_index)
    check(lib().mi355_ivf_residuals(C.byref(desc), _ptr(vectors), C.c_uint64(n), _ptr(centroids), _ptr(out),
                                    _ptr(assign)))
    return out, assign


def _copy_device(dst, src):
    from ._hip import DeviceArray
    if isinstance(dst, DeviceArray):
        dst.copy_from(src)
    else:
        dst.copy_(src)
