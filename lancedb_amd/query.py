"""Host-side mirror of the reference's vector-query interface for this path.

Same names, argument meaning and error behaviour as
`lancedb::Table::vector_search` / `Query::nearest_to` / `VectorQuery`
(/root/reference/rust/lancedb/src/table.rs:1654-1656, query.rs:1011-1021,
:1135-1370) so the parity tests read like the reference's own tests.  The
request is executed by the MI355X engine through the C ABI; there is no CPU
execution path here.
"""
import copy
import time
from dataclasses import dataclass, field
from typing import Any, Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import _abi
from ._lib import InvalidInput, QueryTimeout
from .index import FlatIndex, IvfPqIndex

DEFAULT_TOP_K = 10  # rust/lancedb/src/query.rs:36


@dataclass
class QueryExecutionOptions:
    """rust/lancedb/src/query.rs:626-658: at most `max_batch_length` rows per delivered batch
    (0 = no limit; default 1024) and an optional `timeout` in seconds.  The timeout becomes the
    engine's device-side deadline (mi355_search_params.timeout_ms) and is also checked while the
    batches are handed out (utils/mod.rs:328-392 TimeoutStream)."""
    max_batch_length: int = 1024
    timeout: Optional[float] = None


@dataclass
class VectorQueryRequest:
    """rust/lancedb/src/query.rs:1066-1114 + the QueryRequest base (:818-907).  Fields that act
    on other table columns (`select`, `order_by` on user columns, `filter` as SQL) are carried
    for the table layer; this path produces `_rowid` / `_distance`."""
    limit: Optional[int] = None
    offset: Optional[int] = None
    with_row_id: bool = False
    # ---- QueryRequest base, carried (query.rs:818-907) ----
    select: Optional[Sequence[str]] = None       # Select::All when None; names among the produced columns are honoured
    fast_search: bool = False                    # only indexed data (this engine only ever sees indexed data)
    order_by: Optional[Sequence[Tuple[str, bool]]] = None  # [(column, ascending)] over the produced columns
    norm: Optional[str] = None                   # hybrid-search normalisation: consumer of _distance, not on this path
    reranker: Any = None
    disable_scoring_autoprojection: bool = False
    use_lsm: Optional[bool] = None               # MemWAL routing: forces local execution (table/query.rs:91-105)
    # ---- VectorQueryRequest ----
    approx_mode: Optional[str] = None            # 'fast' | 'normal' | 'accurate' (lib.rs:298-313; RQ indexes only)
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
        """A vector query needs a positive limit (python/python/lancedb/query.py:1176-1178; the Rust
        setter takes a usize, rust/lancedb/src/query.rs:818-907)."""
        if limit is None or int(limit) <= 0:
            raise InvalidInput(1, "Limit is required for ANN/KNN queries")
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

    def select(self, columns):  # query.rs:550-560 (Select::Columns)
        q = self._clone()
        q.request.select = list(columns)
        return q

    def fast_search(self):  # query.rs:509-520
        q = self._clone()
        q.request.fast_search = True
        return q

    def order_by(self, ordering):  # query.rs:618-621: [(column, ascending)] or None
        q = self._clone()
        q.request.order_by = None if ordering is None else [(str(c), bool(a)) for c, a in ordering]
        return q

    def norm(self, norm):  # query.rs:613-616
        q = self._clone()
        q.request.norm = norm
        return q

    def use_lsm(self, enable):  # query.rs:623-626
        q = self._clone()
        q.request.use_lsm = bool(enable)
        return q

    def approx_mode(self, approx_mode):  # query.rs:1351-1357; parsing lib.rs:343-357
        if not isinstance(approx_mode, str) or approx_mode.lower() not in _abi.APPROX_NAMES:
            raise InvalidInput(1, f"approx_mode must be one of 'fast', 'normal', or 'accurate', got '{approx_mode}'")
        q = self._clone()
        q.request.approx_mode = approx_mode.lower()
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
    def execute(self, options: Optional[QueryExecutionOptions] = None):
        """-> dict of numpy columns {_rowid u64, _distance f32[, query_index i32]},
        rows ordered (_distance, _rowid) per query vector
        (table/query.rs:131-381; multi-vector adds `query_index`): the collected stream."""
        if options is None:
            return self._table._execute_vector_query(self.request)
        batches = list(self.execute_with_options(options))
        if len(batches) == 1:
            return batches[0]
        return {k: np.concatenate([b[k] for b in batches]) for k in batches[0]}

    def execute_with_options(self, options: Optional[QueryExecutionOptions] = None):
        """The record-batch stream of the reference (query.rs:1446-1459, :1481-1510): an iterator of
        column dicts of at most `max_batch_length` rows each, stopped by `timeout`."""
        return self._table.query_stream(self.request, options or QueryExecutionOptions())

    def create_plan(self, options: Optional[QueryExecutionOptions] = None):
        return self._table.create_plan(self.request, options or QueryExecutionOptions())

    def explain_plan(self, verbose=False):  # query.rs:1495-1498
        return self._table.create_plan(self.request, QueryExecutionOptions()).explain(verbose)

    def analyze_plan(self, options: Optional[QueryExecutionOptions] = None):
        """Run the query and return its plan with runtime metrics (query.rs:1500-1510 -> table/query.rs:105-112;
        sample python/python/lancedb/query.py:1414-1440): the engine's per-stage device times and counters
        (mi355_last_stats / mi355_flat_last_stats) rendered on the plan nodes they stand for."""
        return self._table.analyze_plan(self.request, options or QueryExecutionOptions())

    to_arrays = execute

    def metric(self, metric):
        """Alias of distance_type (python/python/lancedb/query.py:1596-1612)."""
        return self.distance_type(metric)

    def to_query_object(self):
        """The request this builder stands for, detached (python/python/lancedb/query.py:1790-1822: "can be used to
        serialize a query" — `wire.request_to_json` is that serialisation)."""
        return copy.deepcopy(self.request)

    def output_schema(self):
        """pyarrow schema of the result without running the query (python/python/lancedb/query.py:1763-1769)."""
        import pyarrow as pa
        types = {"_rowid": pa.uint64(), "_distance": pa.float32(), "query_index": pa.int32()}
        return pa.schema([(n, types[n]) for n in self.create_plan().output_columns()])

    # ---- the Python binding's collected forms (python/python/lancedb/query.py:986-1117, :1771-1790) ----------
    def to_batches(self, batch_size=None, *, timeout=None):
        """-> pyarrow.RecordBatchReader over the stream (query.py:1077-1101, :1824): at most `batch_size` rows per
        batch (None = the execution options' default, 1024), `timeout` a timedelta or seconds."""
        import pyarrow as pa
        opts = QueryExecutionOptions()
        if batch_size is not None:
            if int(batch_size) <= 0:
                raise InvalidInput(1, "batch_size must be positive")
            opts.max_batch_length = int(batch_size)
        if timeout is not None:
            opts.timeout = timeout.total_seconds() if hasattr(timeout, "total_seconds") else float(timeout)
        types = {"_rowid": pa.uint64(), "_distance": pa.float32(), "query_index": pa.int32()}
        names = self.create_plan(opts).output_columns()
        schema = pa.schema([(n, types[n]) for n in names])
        stream = self.execute_with_options(opts)

        def gen():
            for b in stream:
                yield pa.record_batch([pa.array(b[n], type=types[n]) for n in names], schema=schema)

        return pa.RecordBatchReader.from_batches(schema, gen())

    def to_arrow(self, *, timeout=None):
        """-> pyarrow.Table (query.py:1771-1790: `to_batches(timeout=...).read_all()`)."""
        return self.to_batches(timeout=timeout).read_all()

    def to_list(self, *, timeout=None):
        """-> list of row dicts (query.py:1103-1117)."""
        return self.to_arrow(timeout=timeout).to_pylist()

    def to_pandas(self, *, timeout=None):
        """-> pandas.DataFrame (query.py:997-1057; the flatten / blob options act on user columns, which this path
        does not produce)."""
        return self.to_arrow(timeout=timeout).to_pandas()

    def to_df(self):
        """Deprecated spelling of to_pandas (query.py:986-995)."""
        return self.to_pandas()


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
        if len(vecs) == 0:  # ensure_vector_query, python/python/lancedb/query.py:332-350
            raise InvalidInput(1, "Vector query must be a non-empty list")
        for v in vecs:
            v = _to_query_vector(v)
            if v.shape[0] == 0:
                raise InvalidInput(1, "Vector query must be a non-empty list")
            if v.shape[0] != self.dim:
                raise InvalidInput(1, f"query vector has dimension {v.shape[0]} but the column has {self.dim}")
            q.request.query_vector.append(v)
        if q.request.limit is None:
            q.request.limit = DEFAULT_TOP_K
        return q

    # ---- BaseTable::create_plan / query (table.rs:566-576, table/query.rs:131-328) ----------
    def create_plan(self, req: VectorQueryRequest, options: Optional[QueryExecutionOptions] = None):
        return create_plan(self, req, options or QueryExecutionOptions())

    def query_stream(self, req: VectorQueryRequest, options: Optional[QueryExecutionOptions] = None):
        return execute_query(self, req, options or QueryExecutionOptions())

    def analyze_plan(self, req: VectorQueryRequest, options: Optional[QueryExecutionOptions] = None):
        return analyze_query_plan(self, req, options or QueryExecutionOptions())

    def _execute_vector_query(self, req: VectorQueryRequest):
        batches = list(execute_query(self, req, QueryExecutionOptions(max_batch_length=0)))
        return batches[0] if len(batches) == 1 else {k: np.concatenate([b[k] for b in batches]) for k in batches[0]}


@dataclass
class VectorPlan:
    """What `table::query::create_plan` (table/query.rs:131-328) decides, as data: the engine runs
    a fixed launch sequence instead of a DataFusion plan, `explain()` names the stages after the
    plan nodes they replace (the reference's tests look for these names: table/query.rs:1074-1165,
    python/python/tests/test_query.py:1229-1245)."""
    table: Any
    request: VectorQueryRequest
    options: QueryExecutionOptions
    k: int                      # limit + offset (table/query.rs:231)
    use_index: bool
    prefilter: bool
    params: Any                 # mi355_search_params
    queries: np.ndarray         # [n, dim] f32

    def explain(self, verbose=False, metrics=None):
        """`metrics` (analyze_plan): {"elapsed_s", "rows", "stats"} of one execution — every node then carries
        `elapsed=..., metrics=[...]` like DataFusion's AnalyzeExec rendering (python/python/lancedb/query.py:1414-1440)."""
        r, lines = self.request, []
        st = (metrics or {}).get("stats") or {}

        def us(*names):  # device time of the stages a node stands for
            return sum(float(st.get("us_" + n, 0.0)) for n in names)

        def m(elapsed_us=None, **kv):
            if metrics is None:
                return ""
            parts = [f"{k2}={v}" for k2, v in kv.items() if v is not None]
            if elapsed_us is not None:
                parts.append(f"elapsed_compute={elapsed_us:.1f}us")
            head = f", elapsed={elapsed_us:.1f}us" if elapsed_us is not None else ""
            return f"{head}, metrics=[{', '.join(parts)}]"

        rows = (metrics or {}).get("rows")
        nq = len(self.queries)
        lim = f"fetch={self.k - (r.offset or 0)}" + (f", skip={r.offset}" if r.offset else "")
        lines.append(f"ProjectionExec: expr=[{', '.join(self.output_columns())}]" + m(output_rows=rows))
        if r.offset:
            lines.append(f"  GlobalLimitExec: {lim}" + m(output_rows=rows))
        if r.order_by:
            lines.append("  SortExec: expr=[" + ", ".join(f"{c} {'ASC' if a else 'DESC'}" for c, a in r.order_by) + "]" + m(output_rows=rows))
        if (r.allow_rowids is not None or r.block_rowids is not None) and not self.prefilter:
            lines.append("  FilterExec: postfilter on _rowid" + m(output_rows=rows))
        if self.use_index:
            kk = self.k * (r.refine_factor or 1)
            if r.refine_factor:
                lines.append(f"  SortExec: TopK(fetch={self.k}), expr=[_distance ASC NULLS LAST, _rowid ASC NULLS LAST]" + m(output_rows=rows))
                lines.append("    KNNVectorDistance: refine, metric=" + (r.distance_type or "index")
                             + m(us("refine"), rows_reranked=nq * kk if metrics is not None else None))
                lines.append("      Take: raw vectors of k * refine_factor = %d rows" % kk)
            lines.append(f"  SortExec: TopK(fetch={kk}), expr=[_distance ASC NULLS LAST, _rowid ASC NULLS LAST]"
                         + m(us("merge"), candidate_lists=st.get("work_items")))
            lines.append(f"    ANNSubIndex: name=mi355_ivf_pq, k={kk}, deltas=1"
                         + (", prefilter=rowid mask" if self.prefilter and (r.allow_rowids is not None or r.block_rowids is not None) else "")
                         + m(us("scan"), rows_scanned=st.get("vectors_scanned"), bytes_read=st.get("code_bytes_scanned"),
                             work_items=st.get("work_items"), scan_variant=st.get("scan_variant"), timed_out=st.get("timed_out")))
            lines.append(f"      ANNIvfPartition: uuid=mi355, minimum_nprobes={r.minimum_nprobes}, "
                         f"maximum_nprobes={r.maximum_nprobes}, deltas=1"
                         + m(us("coarse", "select"), partitions_ranked=st.get("partitions_probed"), queries=st.get("n_queries")))
        else:
            lines.append(f"  SortExec: TopK(fetch={self.k}), expr=[_distance ASC NULLS LAST, _rowid ASC NULLS LAST]"
                         + m(float(st.get("us_rest", 0.0)) if "us_rest" in st else None, fallback_queries=st.get("fallback_queries")))
            lines.append("    KNNVectorDistance: metric=" + (r.distance_type or "l2")
                         + m(float(st.get("us_gemm", 0.0)) if "us_gemm" in st else None, flops=st.get("gemm_flops"),
                             gemm_variant=st.get("gemm_variant")))
            lines.append("      LanceRead: raw column resident in HBM")
        if len(self.queries) > 1:  # create_multi_vector_plan (table/query.rs:334-381)
            lines = ["UnionExec / query_index: %d query vectors in ONE device batch" % len(self.queries)] + ["  " + ln for ln in lines]
        if metrics is not None:
            lines = [f"AnalyzeExec verbose=true, elapsed={metrics.get('elapsed_s', 0.0) * 1e6:.1f}us, metrics=[output_rows={rows}]"] + \
                    ["  " + ln for ln in lines]
        if verbose:
            lines.append(f"-- engine: k={self.params.k} nprobe=[{self.params.nprobe_min},{self.params.nprobe_max or 'all'}] "
                         f"refine_factor={self.params.refine_factor} timeout_ms={self.params.timeout_ms}")
        return "\n".join(lines)

    def output_columns(self):
        cols = ["_rowid", "_distance"] + (["query_index"] if len(self.queries) > 1 else [])
        sel = self.request.select
        if sel is not None:
            unknown = [c for c in sel if c not in cols]
            if unknown:
                raise InvalidInput(1, f"columns {unknown} are not produced by the vector-search path "
                                      "(user columns are taken by the table layer from _rowid)")
            cols = [c for c in cols if c in sel]
        return cols


def create_plan(table, req: VectorQueryRequest, options: QueryExecutionOptions) -> VectorPlan:
    """table::query::create_plan (table/query.rs:131-328) for the vector branch."""
    if not req.query_vector:
        raise InvalidInput(1, "no query vector")
    limit = DEFAULT_TOP_K if req.limit is None else req.limit
    offset = req.offset or 0
    k = limit + offset  # table/query.rs:231
    metric = _abi.METRIC_DEFAULT if req.distance_type is None else _abi.METRIC_NAMES[req.distance_type]
    filtered = req.allow_rowids is not None or req.block_rowids is not None
    pre = filtered and req.prefilter
    timeout_ms = 0 if options.timeout is None else max(1, int(round(_seconds(options.timeout) * 1000)))
    params = _abi.make_params(
        k=k, nprobe_min=req.minimum_nprobes, nprobe_max=req.maximum_nprobes,
        refine_factor=req.refine_factor or 0, metric=metric,
        lower_bound=req.lower_bound, upper_bound=req.upper_bound, timeout_ms=timeout_ms, approx_mode=req.approx_mode,
        allow_rowids=req.allow_rowids if pre else None, block_rowids=req.block_rowids if pre else None)
    use_index = req.use_index and table.index is not None
    if not use_index and table.flat is None:
        raise InvalidInput(1, "bypass_vector_index needs the raw column on the device")
    plan = VectorPlan(table, req, options, k, use_index, pre, params, np.stack(req.query_vector))
    plan.output_columns()  # validates `select`
    return plan


def requires_local_execution(req: VectorQueryRequest) -> bool:
    """table/query.rs:91-105: a query carrying a field the push-down request cannot express must not be
    pushed down — approx_mode / use_lsm in the reference; here also an evaluated row-id mask (the wire
    carries a SQL `filter`, which this layer never sees)."""
    return (req.use_lsm is not None or req.approx_mode is not None or req.allow_rowids is not None
            or req.block_rowids is not None)


def execute_query(table, req: VectorQueryRequest, options: QueryExecutionOptions):
    """table::query::execute_query (table/query.rs:51-65): push the query down when the table has a
    push-down endpoint and the query allows it, otherwise execute locally on this process's GPU
    (execute_generic_query, :115-129).  Returns the batch iterator either way."""
    pushdown = getattr(table, "pushdown", None)
    if pushdown is not None and not requires_local_execution(req):
        from . import wire
        t0 = time.monotonic()  # the deadline covers the round trip (the client's request timeout, remote/table.rs:697)
        body = wire.request_to_json(req)
        # whether the endpoint takes a timeout is read from its signature — NOT by catching TypeError around the
        # call, which would also swallow a TypeError raised inside the endpoint and run the request twice
        import inspect
        try:
            prm = inspect.signature(pushdown).parameters
            takes_timeout = "timeout" in prm or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in prm.values())
        except (TypeError, ValueError):
            takes_timeout = False
        data = pushdown(body, timeout=options.timeout) if takes_timeout else pushdown(body)
        cols = wire.response_from_ipc(data)
        # the server already applied columns / order_by (they travel in the body); a server that predates them returns
        # everything in (_distance, _rowid) order, so apply them here as well — but only what the response still
        # allows: a sort key the server already projected away means the server sorted before projecting
        if req.order_by and all(col in cols for col, _ in req.order_by):
            cols = _order_by(cols, req.order_by)
        if req.select is not None:
            import dataclasses
            cols = _order_and_project(cols, dataclasses.replace(req, order_by=None))
        return _batches(cols, options, t0)
    return execute_generic_query(table, req, options)


def _order_by(out, order_by):
    """Stable multi-key sort over the produced columns: last key first; a DESCENDING key sorts its
    ranks negated, so ties of that key keep the order the less significant keys gave them."""
    order = np.arange(len(next(iter(out.values()))) if out else 0)
    for col, asc in reversed(list(order_by)):
        if col not in out:
            raise InvalidInput(1, f"cannot order by {col!r}: not produced by the vector-search path")
        keys = out[col][order]
        if asc:
            idx = np.argsort(keys, kind="stable")
        else:
            _, rank = np.unique(keys, return_inverse=True)
            idx = np.argsort(-rank.astype(np.int64), kind="stable")
        order = order[idx]
    return {k2: v[order] for k2, v in out.items()}


def _order_and_project(out, req: VectorQueryRequest):
    if req.order_by:
        out = _order_by(out, req.order_by)
    if req.select is not None:
        unknown = [c for c in req.select if c not in out]
        if unknown:
            raise InvalidInput(1, f"columns {unknown} are not produced by the vector-search path "
                                  "(user columns are taken by the table layer from _rowid)")
        out = {c: v for c, v in out.items() if c in req.select}
    return out


def analyze_query_plan(table, req: VectorQueryRequest, options: QueryExecutionOptions) -> str:
    """table::query::analyze_query_plan (table/query.rs:105-112): create the plan, execute it locally with the
    handle's per-stage timers on, render plan + metrics."""
    plan = create_plan(table, req, options)
    handle = table.index if plan.use_index else table.flat
    profiled = False
    try:
        if hasattr(handle, "configure"):
            handle.configure(profile=1) if plan.use_index else handle.configure(profile=True)
            profiled = True
    except TypeError:
        pass
    t0 = time.perf_counter()
    rows = 0
    for batch in execute_generic_query(table, req, options):
        rows += len(next(iter(batch.values()))) if batch else 0
    elapsed = time.perf_counter() - t0
    stats = handle.stats() if hasattr(handle, "stats") else {}
    if profiled:
        handle.configure(profile=0) if plan.use_index else handle.configure(profile=False)
    return plan.explain(verbose=True, metrics={"elapsed_s": elapsed, "rows": rows, "stats": stats})


def execute_generic_query(table, req: VectorQueryRequest, options: QueryExecutionOptions):
    t0 = time.monotonic()
    plan = create_plan(table, req, options)
    q, offset = plan.queries, req.offset or 0
    res = table.index.search(q, plan.params) if plan.use_index else table.flat.search(q, plan.params)
    filtered = req.allow_rowids is not None or req.block_rowids is not None
    rid, dist, qidx = [], [], []
    for i in range(q.shape[0]):
        n = int(res.counts[i])
        r, d = res.rowids[i, :n], res.distances[i, :n]
        if filtered and not plan.prefilter:  # postfilter: the predicate thins out the k results
            keep = np.isin(r, req.allow_rowids) if req.allow_rowids is not None else ~np.isin(r, req.block_rowids)
            r, d = r[keep], d[keep]
        rid.append(r[offset:])
        dist.append(d[offset:])
        qidx.append(np.full(max(len(r) - offset, 0), i, dtype=np.int32))
    out = {"_rowid": np.concatenate(rid), "_distance": np.concatenate(dist)}
    if q.shape[0] > 1:
        out["query_index"] = np.concatenate(qidx)
    if req.order_by:
        out = _order_by(out, req.order_by)
    out = {c: out[c] for c in plan.output_columns()}
    return _batches(out, options, t0)


def _seconds(timeout):
    """QueryExecutionOptions.timeout: seconds, or a datetime.timedelta as the reference's Python API
    passes it (python/python/tests/test_query.py:1846-1873; a zero timeout fails every query)."""
    return timeout.total_seconds() if hasattr(timeout, "total_seconds") else float(timeout)


def _batches(cols, options: QueryExecutionOptions, t0):
    """MaxBatchLengthStream + TimeoutStream (utils/mod.rs:328-471): slices of at most
    max_batch_length rows (0 = one batch); the deadline is checked as each batch is handed out."""
    n = len(next(iter(cols.values()))) if cols else 0
    step = options.max_batch_length or max(n, 1)
    off = 0
    while True:
        if options.timeout is not None and time.monotonic() - t0 > _seconds(options.timeout):
            raise QueryTimeout(3, f"Query timeout: {_seconds(options.timeout)} s")
        yield {c: v[off:off + step] for c, v in cols.items()}
        off += step
        if off >= n:
            return


# ---- utils/mod.rs:151-198, :289-298 --------------------------------------------------------
def supported_vector_data_type(dtype) -> bool:
    """A vector column is a FixedSizeList of any float type or of uint8, or a List of those
    (pyarrow DataType in, as the reference takes an arrow DataType)."""
    import pyarrow as pa
    if pa.types.is_fixed_size_list(dtype):
        return pa.types.is_floating(dtype.value_type) or dtype.value_type == pa.uint8()
    if pa.types.is_list(dtype) or pa.types.is_large_list(dtype):
        return supported_vector_data_type(dtype.value_type)
    return False


def infer_vector_dim(dtype) -> int:
    """lance::index::vector::utils::infer_vector_dim [EXT]: the list size of a vector column, or of
    the vectors inside a multivector (List<FixedSizeList>) column."""
    import pyarrow as pa
    if pa.types.is_fixed_size_list(dtype) and (pa.types.is_floating(dtype.value_type) or dtype.value_type == pa.uint8()):
        return dtype.list_size
    if (pa.types.is_list(dtype) or pa.types.is_large_list(dtype)) and pa.types.is_fixed_size_list(dtype.value_type):
        return infer_vector_dim(dtype.value_type)
    raise InvalidInput(1, f"Data type is not a vector (FixedSizeListArray or List<FixedSizeListArray>), but {dtype}")


def default_vector_column(schema, dim: Optional[int] = None) -> str:
    """utils/mod.rs:151-198: the one vector column (of dimension `dim`, if given) of an arrow
    schema, looking inside structs; none -> InvalidInput, several -> a schema error."""
    import pyarrow as pa
    candidates = []

    def collect(fld, path):
        path = path + [fld.name]
        try:
            d = infer_vector_dim(fld.type)
            if dim is None or dim == d:
                candidates.append(".".join(path))
                return
        except InvalidInput:
            pass
        if pa.types.is_struct(fld.type):
            for child in fld.type:
                collect(child, path)

    for f in schema:
        collect(f, [])
    if not candidates:
        raise InvalidInput(1, f"No vector column found to match with the query vector dimension: {dim or 0}")
    if len(candidates) != 1:
        raise InvalidInput(1, "More than one vector columns found, please specify which column to create index or "
                              f"query: {candidates}")
    return candidates[0]
