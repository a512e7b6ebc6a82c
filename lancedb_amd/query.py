"""Host-side mirror of the reference's vector-query interface for this path.

Same names, argument meaning and error behaviour as
`lancedb::Table::vector_search` / `Query::nearest_to` / `VectorQuery`
(/root/reference/rust/lancedb/src/table.rs:1654-1656, query.rs:1011-1021,
:1135-1370) so the parity tests read like the reference's own tests.  The
request is executed by the MI355X engine through the C ABI; there is no CPU
execution path here.
"""
import copy
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _abi
from ._lib import InvalidInput
from .index import FlatIndex, IvfPqIndex

DEFAULT_TOP_K = 10  # rust/lancedb/src/query.rs:36


@dataclass
class VectorQueryRequest:
    """rust/lancedb/src/query.rs:1066-1114 (numeric subset; filters/projection
    belong to the table layer and are out of scope)."""
    limit: Optional[int] = None
    offset: Optional[int] = None
    with_row_id: bool = False
    column: Optional[str] = None
    query_vector: List[np.ndarray] = field(default_factory=list)
    minimum_nprobes: int = 20
    maximum_nprobes: Optional[int] = 20
    lower_bound: Optional[float] = None
    upper_bound: Optional[float] = None
    ef: Optional[int] = None
    refine_factor: Optional[int] = None
    distance_type: Optional[str] = None
    use_index: bool = True
    # QueryRequest.filter / prefilter (query.rs:489-507, :899).  The predicate itself is
    # evaluated by the table layer (out of scope); what reaches the ANN nodes is the set of
    # permitted / dropped _rowids (lance RowIdMask [EXT]).
    allow_rowids: Optional[np.ndarray] = None
    block_rowids: Optional[np.ndarray] = None
    prefilter: bool = True  # default in the reference; False = postfilter (query.rs:496-507)


def _to_query_vector(v):
    """IntoQueryVector: everything is cast to Float32 (query.rs:131-373, :1013)."""
    a = np.asarray(v)
    if a.dtype == object or a.ndim != 1:
        raise InvalidInput(1, "query vector must be a 1-D numeric array")
    return np.ascontiguousarray(a, dtype=np.float32)


class VectorQuery:
    """Builder with the reference's setters; `execute()` runs on the GPU."""

    def __init__(self, table, request=None):
        self._table = table
        self.request = request or VectorQueryRequest()

    def _clone(self):
        return VectorQuery(self._table, copy.deepcopy(self.request))

    # ---- QueryBase -----------------------------------------------------
    def limit(self, limit):
        q = self._clone()
        q.request.limit = int(limit)
        return q

    def offset(self, offset):
        q = self._clone()
        q.request.offset = int(offset)
        return q

    def with_row_id(self):
        q = self._clone()
        q.request.with_row_id = True
        return q

    # ---- VectorQuery ---------------------------------------------------
    def column(self, column):
        q = self._clone()
        q.request.column = column
        return q

    def add_query_vector(self, vector):  # query.rs:1185-1189
        q = self._clone()
        q.request.query_vector.append(_to_query_vector(vector))
        return q

    def nprobes(self, nprobes):  # query.rs:1216-1220: sets both bounds
        q = self._clone()
        q.request.minimum_nprobes = int(nprobes)
        q.request.maximum_nprobes = int(nprobes)
        return q

    def minimum_nprobes(self, minimum_nprobes):  # query.rs:1232-1249
        if minimum_nprobes == 0:
            raise InvalidInput(1, "minimum_nprobes must be greater than 0")
        mx = self.request.maximum_nprobes
        if mx is not None and minimum_nprobes > mx:
            raise InvalidInput(1, "minimum_nprobes must be less than or equal to maximum_nprobes")
        q = self._clone()
        q.request.minimum_nprobes = int(minimum_nprobes)
        return q

    def maximum_nprobes(self, maximum_nprobes):  # query.rs:1265-1280
        if maximum_nprobes is not None:
            if maximum_nprobes == 0:
                raise InvalidInput(1, "maximum_nprobes must be greater than 0")
            if maximum_nprobes < self.request.minimum_nprobes:
                raise InvalidInput(1, "maximum_nprobes must be greater than or equal to minimum_nprobes")
        q = self._clone()
        q.request.maximum_nprobes = None if maximum_nprobes is None else int(maximum_nprobes)
        return q

    def distance_range(self, lower_bound=None, upper_bound=None):  # query.rs:1284-1288
        q = self._clone()
        q.request.lower_bound, q.request.upper_bound = lower_bound, upper_bound
        return q

    def ef(self, ef):  # HNSW only; carried, unused by IVF-PQ
        q = self._clone()
        q.request.ef = int(ef)
        return q

    def refine_factor(self, refine_factor):  # query.rs:1329-1332
        q = self._clone()
        q.request.refine_factor = int(refine_factor)
        return q

    def distance_type(self, distance_type):  # query.rs:1347-1350
        if isinstance(distance_type, str):
            distance_type = distance_type.lower()
        if distance_type not in _abi.METRIC_NAMES:
            raise InvalidInput(1, f"unknown distance type {distance_type!r}")
        q = self._clone()
        q.request.distance_type = distance_type
        return q

    def only_if_rowids(self, allow=None, block=None):
        """The evaluated form of `only_if(filter)` (query.rs:440-455): the _rowids the
        predicate keeps (`allow`) or drops (`block`)."""
        if (allow is None) == (block is None):
            raise InvalidInput(1, "give exactly one of allow / block")
        q = self._clone()
        q.request.allow_rowids = None if allow is None else np.unique(np.asarray(allow, dtype=np.uint64))
        q.request.block_rowids = None if block is None else np.unique(np.asarray(block, dtype=np.uint64))
        return q

    def postfilter(self):  # query.rs:496-507: filter applied to the k results of the search
        q = self._clone()
        q.request.prefilter = False
        return q

    def bypass_vector_index(self):  # query.rs:1367-1370
        q = self._clone()
        q.request.use_index = False
        return q

    # ---- ExecutableQuery -------------------------------------------------
    def execute(self):
        """-> dict of numpy columns {_rowid u64, _distance f32[, query_index i32]},
        rows ordered (_distance, _rowid) per query vector
        (table/query.rs:131-381; multi-vector adds `query_index`)."""
        return self._table._execute_vector_query(self.request)

    to_arrays = execute


class VectorTable:
    """The slice of `BaseTable` this path needs (table.rs:549-576): holds the
    device-resident index and/or raw column and executes VectorQueryRequests."""

    def __init__(self, index: Optional[IvfPqIndex] = None, flat: Optional[FlatIndex] = None,
                 index_metric: Optional[str] = None):
        if index is None and flat is None:
            raise InvalidInput(1, "VectorTable needs an IvfPqIndex and/or a FlatIndex")
        self.index, self.flat = index, flat
        self.dim = index.dim if index is not None else flat.dim

    def query(self):
        return VectorQuery(self)

    def vector_search(self, vector):  # table.rs:1654-1656
        return self.query_nearest_to(vector)

    search = vector_search

    def query_nearest_to(self, vector):  # Query::nearest_to, query.rs:1011-1021
        q = VectorQuery(self)
        a = np.asarray(vector)
        vecs = [a] if a.ndim == 1 else list(a)  # a list of vectors = multi-vector query
        for v in vecs:
            v = _to_query_vector(v)
            if v.shape[0] != self.dim:
                raise InvalidInput(1, f"query vector has dimension {v.shape[0]} but the column has {self.dim}")
            q.request.query_vector.append(v)
        if q.request.limit is None:
            q.request.limit = DEFAULT_TOP_K
        return q

    def _execute_vector_query(self, req: VectorQueryRequest):
        if not req.query_vector:
            raise InvalidInput(1, "no query vector")
        limit = DEFAULT_TOP_K if req.limit is None else req.limit
        offset = req.offset or 0
        k = limit + offset  # table/query.rs:231
        metric = _abi.METRIC_DEFAULT if req.distance_type is None else _abi.METRIC_NAMES[req.distance_type]
        filtered = req.allow_rowids is not None or req.block_rowids is not None
        pre = filtered and req.prefilter
        params = _abi.make_params(
            k=k, nprobe_min=req.minimum_nprobes, nprobe_max=req.maximum_nprobes,
            refine_factor=req.refine_factor or 0, metric=metric,
            lower_bound=req.lower_bound, upper_bound=req.upper_bound,
            allow_rowids=req.allow_rowids if pre else None, block_rowids=req.block_rowids if pre else None)
        q = np.stack(req.query_vector)
        use_index = req.use_index and self.index is not None
        if use_index:
            res = self.index.search(q, params)
        else:
            if self.flat is None:
                raise InvalidInput(1, "bypass_vector_index needs the raw column on the device")
            res = self.flat.search(q, params)
        rid, dist, qidx = [], [], []
        for i in range(q.shape[0]):
            n = int(res.counts[i])
            r, d = res.rowids[i, :n], res.distances[i, :n]
            if filtered and not pre:  # postfilter: the predicate thins out the k results
                keep = np.isin(r, req.allow_rowids) if req.allow_rowids is not None else ~np.isin(r, req.block_rowids)
                r, d = r[keep], d[keep]
            rid.append(r[offset:])
            dist.append(d[offset:])
            qidx.append(np.full(max(len(r) - offset, 0), i, dtype=np.int32))
        out = {"_rowid": np.concatenate(rid), "_distance": np.concatenate(dist)}
        if q.shape[0] > 1:
            out["query_index"] = np.concatenate(qidx)
        return out
