"""Multi-GPU search: one process per GPU, IVF partitions sharded across ranks.

SURVEY.md §8e: a query's result is the top-k of the union of its probed
partitions, and partitions are scanned independently, so the partition list
shards with no data-path exchange until the very end: every rank scans the
probed partitions it owns (`IvfPqIndex(shard_count=world, shard_rank=rank)`,
the deterministic `mi355_shard_plan`), then ONE packed all-gather of the per-rank
candidate records (RCCL over xGMI, 160 B per query per rank at k = 10:
latency-bound) and a k-way merge on every rank.

The whole exchange lives behind the C ABI (`mi355_comm_*`,
`mi355_search_sharded` in include/mi355_ann.h, csrc/ann_comm.hip): this module
is a ctypes veneer, so a Rust host needs nothing from Python.
The 128-byte communicator id travels over whatever channel the host already has;
`exchange_id_via_file` is the dependency-free option used by the launcher test.
The reference has no collective at all, this exchange is the
engine's own.
"""
import ctypes as C
import os
import time

from . import _abi
from ._lib import check, lib
from .index import SearchResult, _Handle, _run_search


def coarse_slice(nlist, world, rank):
    """Centroid slice a rank scores in the two-phase search (contiguous, balanced)."""
    lo, hi = C.c_uint32(0), C.c_uint32(0)
    check(lib().mi355_coarse_slice(C.c_uint32(nlist), C.c_uint32(world), C.c_uint32(rank), C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def unique_id():
    """Rank 0: a fresh communicator id (ncclGetUniqueId), 128 bytes."""
    buf = C.create_string_buffer(_abi.COMM_ID_BYTES)
    check(lib().mi355_comm_unique_id(buf))
    return buf.raw


def exchange_id_via_file(path, rank, timeout_s=120.0):
    """Rank 0 writes the id to `path` (atomically), the others wait for it.  `path` must be
    fresh per launch (e.g. carry the launcher's pid): a file left by an earlier run would hand
    the other ranks a dead id, so rank 0 refuses to start over an existing one."""
    if rank == 0:
        if os.path.exists(path):
            raise FileExistsError(f"{path} exists: a communicator id file must be new for every launch")
        uid = unique_id()
        tmp = f"{path}.tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid
    t0 = time.time()
    while time.time() - t0 < timeout_s:
        try:
            with open(path, "rb") as f:
                uid = f.read()
            if len(uid) == _abi.COMM_ID_BYTES:
                return uid
        except OSError:
            pass
        time.sleep(0.02)
    raise TimeoutError(f"no communicator id at {path} after {timeout_s} s")


class Comm(_Handle):
    """One rank's communicator.  `Comm(uid, rank, world)` is RCCL (mi355_comm_create is collective:
    every rank calls it, one process per GPU); `Comm.loopback(world)` returns the `world`
    communicators of a loopback group — all ranks in this process on one device, each driven
    from its own thread (`run_ranks`)."""
    _close_fn = "mi355_comm_destroy"

    def __init__(self, uid, rank, world, device=0, _handle=None):
        super().__init__()
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        if _handle is not None:
            self._h = _handle
            return
        if len(uid) != _abi.COMM_ID_BYTES:
            raise ValueError(f"communicator id must be {_abi.COMM_ID_BYTES} bytes")
        check(lib().mi355_comm_create(C.c_char_p(uid), C.c_uint32(rank), C.c_uint32(world), C.c_int32(device),
                                      C.byref(self._h)))

    @classmethod
    def loopback(cls, world, device=0):
        hs = (C.c_void_p * int(world))()
        check(lib().mi355_comm_create_loopback(C.c_uint32(world), C.c_int32(device), hs))
        return [cls(None, r, world, device, _handle=C.c_void_p(hs[r])) for r in range(int(world))]

    def stats(self):
        """Load report of the last sharded search (identical on every rank): per-rank scanned
        rows, max/mean imbalance, all-gathers issued and bytes received."""
        s = _abi.CommStats()
        s.struct_size = C.sizeof(_abi.CommStats)
        check(lib().mi355_comm_last_stats(self._h, C.byref(s)))
        return {"world": s.world, "rank": s.rank, "n_gathers": s.n_gathers, "bytes_gathered": s.bytes_gathered,
                "rows_scanned": [int(s.rows_scanned[r]) for r in range(s.world)], "imbalance": float(s.imbalance),
                "us_exchange": float(s.us_exchange), "overlapped": bool(s.overlapped)}


def run_ranks(fns):
    """Run one callable per rank of a loopback group, each on its own thread (the collective
    calls rendezvous inside the library; ctypes drops the GIL for their duration).  Returns the
    results in rank order; the first exception of any rank is re-raised."""
    import threading
    out, err = [None] * len(fns), [None] * len(fns)

    def work(i):
        try:
            out[i] = fns[i]()
        except BaseException as e:  # noqa: BLE001 - handed to the caller below
            err[i] = e

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(fns))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in err:
        if e is not None:
            raise e
    return out


class ShardedSearcher:
    """`index` is this rank's shard handle (`IvfPqIndex(shard_count=world, shard_rank=rank)`).
    `search` is collective; its result is identical on every rank and identical to the
    unsharded search — including `refine_factor` (the global k * refine_factor ANN
    candidates are merged first, then every rank refines the ones whose raw vectors it
    owns) and `maximum_nprobes`."""

    def __init__(self, index, comm, shard_coarse=False, overlap=True):
        """shard_coarse: two-phase search (C4, nlist = 65536): every rank scores only its
        slice of the centroids; one extra all-gather of `nprobe` (partition, distance) records
        per query per rank selects the global probe list before the scan.
        overlap: device-I/O calls queue their exchange on the communicator's stream so that
        the next call's scan overlaps it (results are complete after `index.sync()`)."""
        self.index, self.comm = index, comm
        self.flags = (_abi.SHARD_COARSE if shard_coarse else 0) | (0 if overlap else _abi.SHARD_NO_OVERLAP)

    def search(self, queries, params, out=None):
        def fn(handle, q, nq, params_ref, ids, dist, cnt):
            return lib().mi355_search_sharded(handle, self.comm._h, q, nq, params_ref, C.c_uint32(self.flags), ids, dist, cnt)
        return _run_search(fn, self.index._h, self.index.dim, queries, params, out)


class ShardedFlatSearcher:
    """Flat search with the ROWS sharded across ranks (`flat` holds this rank's slice, its
    row_ids the global ids): same gather + merge."""

    def __init__(self, flat, comm):
        self.flat, self.comm = flat, comm

    def search(self, queries, params, out=None):
        def fn(handle, q, nq, params_ref, ids, dist, cnt):
            return lib().mi355_flat_search_sharded(handle, self.comm._h, q, nq, params_ref, ids, dist, cnt)
        return _run_search(fn, self.flat._h, self.flat.dim, queries, params, out)


__all__ = ["Comm", "ShardedSearcher", "ShardedFlatSearcher", "coarse_slice", "unique_id", "exchange_id_via_file", "run_ranks",
           "SearchResult"]
