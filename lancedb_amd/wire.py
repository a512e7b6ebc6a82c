"""The remote wire format of a vector query: JSON request
body <-> VectorQueryRequest, and the Arrow IPC file response.

This is the data format on the caller's side of the hot path when the engine
sits behind a server existing LanceDB clients already speak to.  Field names and
defaults follow the reference's client (rust/lancedb/src/remote/table.rs:724-930;
the body pinned by test_query_vector_default_values, :4640-4699); no HTTP here.
"""
import json

import numpy as np

from ._lib import InvalidInput, NotSupported
from .query import DEFAULT_TOP_K, VectorQuery, VectorQueryRequest

JSON_CONTENT_TYPE = "application/json"
ARROW_FILE_CONTENT_TYPE = "application/vnd.apache.arrow.file"
_ISIZE_MAX = (1 << 63) - 1  # `k` a client sends when no limit was set (remote/table.rs:741-743)


def request_to_json(req: VectorQueryRequest, version=None) -> dict:
    """What the reference's client puts on the wire for `req` (remote/table.rs:833-930)."""
    body = {"prefilter": bool(req.prefilter)}
    if req.offset is not None:
        body["offset"] = int(req.offset)
    body["k"] = _ISIZE_MAX if req.limit is None else int(req.limit)
    if req.select is not None:  # Select::Columns (remote/table.rs:762-771)
        body["columns"] = [str(c) for c in req.select]
    if req.fast_search:  # remote/table.rs:796-798
        body["fast_search"] = True
    if req.with_row_id:
        body["with_row_id"] = True
    if req.order_by:  # remote/table.rs:820-833
        body["order_by"] = [{"column_name": c, "ascending": bool(asc), "nulls_first": False} for c, asc in req.order_by]
    if req.distance_type is not None:
        body["distance_type"] = req.distance_type
    if req.approx_mode is not None:  # remote/table.rs:844-846; body pinned at :4694-4745
        body["approx_mode"] = req.approx_mode
    body["nprobes"] = int(req.minimum_nprobes)
    body["minimum_nprobes"] = int(req.minimum_nprobes)
    body["maximum_nprobes"] = 0 if req.maximum_nprobes is None else int(req.maximum_nprobes)
    body["lower_bound"] = req.lower_bound
    body["upper_bound"] = req.upper_bound
    body["ef"] = req.ef
    body["refine_factor"] = req.refine_factor
    if req.column is not None:
        body["vector_column"] = req.column
    if not req.use_index:
        body["bypass_vector_index"] = True
    vecs = [np.asarray(v, dtype=np.float32) for v in req.query_vector]
    if len(vecs) == 0:
        body["vector"] = []
    elif len(vecs) == 1:
        body["vector"] = [float(x) for x in vecs[0]]  # f32 values widened to f64, as serde does
    else:
        body["vector"] = [[float(x) for x in v] for v in vecs]
    body["version"] = version
    return body


def request_from_json(body) -> VectorQueryRequest:
    """Server side: the JSON body -> the request the engine executes."""
    if isinstance(body, (bytes, bytearray, str)):
        body = json.loads(body)
    if "filter" in body and body["filter"]:
        raise NotSupported(4, "SQL filters are evaluated by the table layer; pass the permitted row ids "
                              "(VectorQuery.only_if_rowids) to the engine")
    req = VectorQueryRequest()
    k = body.get("k")
    req.limit = DEFAULT_TOP_K if k is None else (None if int(k) >= _ISIZE_MAX else int(k))
    if req.limit is None:
        raise InvalidInput(1, "a vector query needs a limit")
    req.offset = body.get("offset")
    req.prefilter = bool(body.get("prefilter", True))
    req.with_row_id = bool(body.get("with_row_id", False))
    cols = body.get("columns")
    if cols is not None:
        if isinstance(cols, dict):  # Select::Dynamic / Select::Expr: SQL expressions belong to the table layer
            raise NotSupported(4, "computed columns are evaluated by the table layer, not by the vector-search path")
        req.select = [str(c) for c in cols]
    req.fast_search = bool(body.get("fast_search", False))
    if body.get("order_by"):
        req.order_by = [(o["column_name"], bool(o.get("ascending", True))) for o in body["order_by"]]
    req.distance_type = body.get("distance_type")
    am = body.get("approx_mode")
    if am is not None:
        from . import _abi
        if not isinstance(am, str) or am.lower() not in _abi.APPROX_NAMES:  # lib.rs:343-357
            raise InvalidInput(1, f"approx_mode must be one of 'fast', 'normal', or 'accurate', got '{am}'")
        req.approx_mode = am.lower()
    # old clients only send `nprobes` (remote/table.rs:846-851)
    req.minimum_nprobes = int(body.get("minimum_nprobes", body.get("nprobes", 20)))
    mx = body.get("maximum_nprobes", body.get("nprobes", req.minimum_nprobes))
    req.maximum_nprobes = None if mx in (0, None) else int(mx)
    req.lower_bound, req.upper_bound = body.get("lower_bound"), body.get("upper_bound")
    req.ef, req.refine_factor = body.get("ef"), body.get("refine_factor")
    req.column = body.get("vector_column")
    req.use_index = not bool(body.get("bypass_vector_index", False))
    vec = body.get("vector", [])
    if len(vec) and isinstance(vec[0], (list, tuple)):
        req.query_vector = [np.asarray(v, dtype=np.float32) for v in vec]
    elif len(vec):
        req.query_vector = [np.asarray(vec, dtype=np.float32)]
    return req


def response_to_ipc(columns: dict, with_row_id=True) -> bytes:
    """Arrow IPC *file* bytes ({_rowid: uint64, _distance: float32[, query_index: int32]}),
    what the reference's client parses (table/query.rs:636-682)."""
    import pyarrow as pa
    arrays, names = [], []
    if with_row_id and "_rowid" in columns:
        arrays.append(pa.array(columns["_rowid"], type=pa.uint64()))
        names.append("_rowid")
    if "_distance" in columns:  # absent only when `columns` projected it away
        arrays.append(pa.array(columns["_distance"], type=pa.float32()))
        names.append("_distance")
    if "query_index" in columns:
        arrays.append(pa.array(columns["query_index"], type=pa.int32()))
        names.append("query_index")
    batch = pa.record_batch(arrays, names=names)
    sink = pa.BufferOutputStream()
    with pa.ipc.new_file(sink, batch.schema) as w:
        w.write_batch(batch)
    return sink.getvalue().to_pybytes()


def response_from_ipc(data: bytes) -> dict:
    """Client side: the Arrow IPC file of a response -> numpy columns (table/query.rs:636-682)."""
    import pyarrow as pa
    t = pa.ipc.open_file(pa.BufferReader(data)).read_all()
    return {name: t.column(name).to_numpy() for name in t.schema.names}


def handle_query(table, body, allow_rowids=None, block_rowids=None):
    """POST /v1/table/<name>/query/ without the HTTP: -> (content type, Arrow IPC file bytes).
    `allow_rowids` / `block_rowids`: the evaluated `filter` of the body, if it had one."""
    if isinstance(body, (bytes, bytearray, str)):
        body = json.loads(body)
    had_filter = bool(body.get("filter"))
    req = request_from_json({k: v for k, v in body.items() if k != "filter"})
    if had_filter and allow_rowids is None and block_rowids is None:
        raise NotSupported(4, "the body carries a filter but no evaluated row ids were supplied")
    if len(req.query_vector) == 0:
        raise InvalidInput(1, "no query vector")
    for v in req.query_vector:
        if v.shape[0] != table.dim:
            raise InvalidInput(1, f"query vector has dimension {v.shape[0]} but the column has {table.dim}")
    q = VectorQuery(table, req)
    if allow_rowids is not None or block_rowids is not None:
        q = q.only_if_rowids(allow=allow_rowids, block=block_rowids)
        q.request.prefilter = req.prefilter
    out = q.execute()
    return ARROW_FILE_CONTENT_TYPE, response_to_ipc(out, with_row_id=True)
