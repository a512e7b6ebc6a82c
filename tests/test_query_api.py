"""Builder semantics of the VectorQuery mirror, written after the reference's
own unit tests (rust/lancedb/src/query.rs:1232-1288, nodejs/__test__/table.test.ts:967-972)."""
import numpy as np
import pytest

import lancedb_amd
from lancedb_amd import VectorQuery, VectorQueryRequest
from lancedb_amd.query import QueryExecutionOptions, VectorTable


class _FakeTable(VectorTable):
    def __init__(self, dim=4):
        self.index, self.flat, self.dim = None, None, dim
        self.seen = None

    def _execute_vector_query(self, req):
        self.seen = req
        return {}


def test_request_defaults():
    r = VectorQueryRequest()  # query.rs:1097-1114
    assert (r.minimum_nprobes, r.maximum_nprobes) == (20, 20)
    assert r.refine_factor is None and r.distance_type is None and r.use_index
    assert r.lower_bound is None and r.upper_bound is None


def test_nearest_to_sets_default_limit_and_casts_to_f32():
    t = _FakeTable()
    q = t.vector_search(np.array([1, 2, 3, 4], dtype=np.float64))
    assert q.request.limit == lancedb_amd.DEFAULT_TOP_K == 10
    assert q.request.query_vector[0].dtype == np.float32
    with pytest.raises(lancedb_amd.InvalidInput, match="dimension"):
        t.vector_search([1.0, 2.0])


def test_nprobes_validation_messages():
    q = _FakeTable().vector_search([0, 0, 0, 0])
    with pytest.raises(lancedb_amd.InvalidInput, match="minimum_nprobes must be greater than 0"):
        q.minimum_nprobes(0)
    with pytest.raises(lancedb_amd.InvalidInput, match="maximum_nprobes must be greater than 0"):
        q.maximum_nprobes(0)
    with pytest.raises(lancedb_amd.InvalidInput, match="less than or equal to maximum_nprobes"):
        q.minimum_nprobes(21)
    with pytest.raises(lancedb_amd.InvalidInput, match="greater than or equal to minimum_nprobes"):
        q.maximum_nprobes(19)
    q2 = q.nprobes(64)
    assert (q2.request.minimum_nprobes, q2.request.maximum_nprobes) == (64, 64)
    assert q.request.minimum_nprobes == 20  # builders are persistent values (Clone semantics)
    q3 = q.maximum_nprobes(None).minimum_nprobes(500)
    assert q3.request.maximum_nprobes is None


def test_setters_carry_through():
    t = _FakeTable()
    q = (t.vector_search([1, 2, 3, 4]).limit(5).offset(2).refine_factor(3).distance_type("Cosine")
         .distance_range(0.1, 0.9).bypass_vector_index().with_row_id().add_query_vector([4, 3, 2, 1]))
    q.execute()
    r = t.seen
    assert (r.limit, r.offset, r.refine_factor, r.distance_type) == (5, 2, 3, "cosine")
    assert (r.lower_bound, r.upper_bound, r.use_index, r.with_row_id) == (0.1, 0.9, False, True)
    assert len(r.query_vector) == 2
    with pytest.raises(lancedb_amd.InvalidInput):
        q.distance_type("hamming-ish")


def test_base_request_fields_and_approx_mode_carry_through():
    """QueryRequest base (query.rs:818-907) and approx_mode (query.rs:1092, :1351-1357; parsing lib.rs:343-357;
    the reference's own checks: query.rs:1720-1756)."""
    t = _FakeTable()
    q = (t.vector_search([1, 2, 3, 4]).select(["_distance"]).fast_search().order_by([("_distance", False)])
         .norm("rank").use_lsm(False).approx_mode("Accurate"))
    q.execute()
    r = t.seen
    assert r.select == ["_distance"] and r.fast_search and r.order_by == [("_distance", False)]
    assert r.norm == "rank" and r.use_lsm is False and r.approx_mode == "accurate"
    assert VectorQueryRequest().approx_mode is None and VectorQueryRequest().prefilter and not VectorQueryRequest().fast_search
    assert t.vector_search([1, 2, 3, 4]).approx_mode("FAST").request.approx_mode == "fast"
    with pytest.raises(lancedb_amd.InvalidInput, match="approx_mode must be one of 'fast', 'normal', or 'accurate', got 'invalid'"):
        t.vector_search([1, 2, 3, 4]).approx_mode("invalid")


class _ArrayIndex:
    """Stands in for the device handles: returns canned per-query rows (the engine itself is GPU-only)."""
    dim = 4

    def __init__(self, n):
        self.n, self.params = n, None

    def search(self, q, params):
        from lancedb_amd.index import SearchResult
        self.params = params
        k = params.k
        ids = np.tile(np.arange(k, dtype=np.uint64), (len(q), 1)) + 100 * np.arange(len(q), dtype=np.uint64)[:, None]
        d = np.tile(np.arange(k, dtype=np.float32), (len(q), 1))
        return SearchResult(ids, d, np.full(len(q), min(k, self.n), np.uint32))


def test_execution_options_slice_batches_and_time_out():
    """max_batch_length (default 1024; query.rs:626-658, utils/mod.rs:395-471) and timeout."""
    from lancedb_amd.query import QueryExecutionOptions
    t = VectorTable(index=_ArrayIndex(5000))
    q = t.vector_search([0, 0, 0, 0]).limit(2500)
    batches = list(q.execute_with_options())
    assert [len(b["_rowid"]) for b in batches] == [1024, 1024, 452]
    assert (np.concatenate([b["_rowid"] for b in batches]) == np.arange(2500)).all()
    assert [len(b["_rowid"]) for b in q.execute_with_options(QueryExecutionOptions(max_batch_length=0))] == [2500]
    assert [len(b["_distance"]) for b in q.execute_with_options(QueryExecutionOptions(max_batch_length=1000))] == [1000, 1000, 500]
    assert len(q.execute()["_rowid"]) == 2500 and len(q.execute(QueryExecutionOptions(max_batch_length=7))["_rowid"]) == 2500
    # the timeout reaches the engine as its device-side deadline and stops the stream
    list(q.execute_with_options(QueryExecutionOptions(timeout=2.5)))
    assert t.index.params.timeout_ms == 2500
    it = q.execute_with_options(QueryExecutionOptions(max_batch_length=10, timeout=0.05))
    next(it)
    import time
    time.sleep(0.08)
    with pytest.raises(lancedb_amd.QueryTimeout, match="Query timeout"):
        next(it)
    # offset, order_by over the produced columns, projection
    r = t.vector_search([0, 0, 0, 0]).limit(5).offset(3).order_by([("_distance", False)]).select(["_rowid"]).execute()
    assert list(r) == ["_rowid"] and r["_rowid"].tolist() == [7, 6, 5, 4, 3]
    with pytest.raises(lancedb_amd.InvalidInput, match="not produced"):
        t.vector_search([0, 0, 0, 0]).select(["title"]).execute()


def test_plan_names_the_nodes_it_replaces():
    """The reference's tests look for these node names (table/query.rs:1074-1165,
    python/python/tests/test_query.py:1229-1245, :1261-1271)."""
    t = VectorTable(index=_ArrayIndex(100), flat=_ArrayIndex(100))
    plan = t.vector_search([0, 0, 0, 0]).nprobes(7).refine_factor(4).explain_plan(True)
    for needle in ("ANNIvfPartition", "minimum_nprobes=7", "ANNSubIndex", "k=40", "KNNVectorDistance", "Take", "TopK(fetch=10)"):
        assert needle in plan, (needle, plan)
    flat = t.vector_search([0, 0, 0, 0]).bypass_vector_index().explain_plan()
    assert "KNNVectorDistance" in flat and "ANNSubIndex" not in flat
    multi = t.vector_search(np.zeros((3, 4))).create_plan()
    assert multi.k == 10 and len(multi.queries) == 3 and "query_index" in multi.explain()


def test_analyze_plan_renders_runtime_metrics_on_the_plan_nodes():
    """query.rs:1500-1510 / table/query.rs:105-112; the reference's sample output (python/python/lancedb/query.py:
    1414-1440) carries `elapsed=` and `metrics=[...]` on every node under an AnalyzeExec root."""
    class _Timed(_ArrayIndex):
        def stats(self):
            return {"us_coarse": 40.0, "us_select": 60.0, "us_scan": 300.0, "us_merge": 20.0, "us_refine": 9.0, "n_queries": 1,
                    "partitions_probed": 7, "vectors_scanned": 12345, "code_bytes_scanned": 12345 * 96, "work_items": 7,
                    "scan_variant": 2, "timed_out": 0}

        def configure(self, **kw):
            self.cfg = kw
    t = VectorTable(index=_Timed(100), flat=_Timed(100))
    text = t.vector_search([0, 0, 0, 0]).nprobes(7).refine_factor(4).limit(5).analyze_plan()
    first = text.splitlines()[0]
    assert first.startswith("AnalyzeExec verbose=true, elapsed=") and "output_rows=5" in first
    for needle in ("ANNIvfPartition", "elapsed=100.0us", "partitions_ranked=7", "ANNSubIndex", "rows_scanned=12345",
                   "bytes_read=1185120", "elapsed_compute=300.0us", "KNNVectorDistance: refine", "rows_reranked=20",
                   "TopK(fetch=20)", "elapsed=20.0us"):
        assert needle in text, (needle, text)
    assert t.index.cfg == {"profile": 0}  # the timers are switched off again
    # explain_plan stays free of metrics
    assert "metrics=" not in t.vector_search([0, 0, 0, 0]).explain_plan()


def test_pushdown_dispatch_follows_the_reference_rules():
    """table/query.rs:51-65, :91-105: queries go to the push-down endpoint unless approx_mode or
    use_lsm is set (the wire request has no field for them)."""
    from lancedb_amd import wire
    t = VectorTable(index=_ArrayIndex(50))
    bodies = []

    def endpoint(body):
        bodies.append(body)
        return wire.response_to_ipc({"_rowid": np.array([9, 8], np.uint64), "_distance": np.array([0.5, 0.75], np.float32)})

    t.pushdown = endpoint
    r = t.vector_search([1, 2, 3, 4]).limit(2).execute()
    assert r["_rowid"].tolist() == [9, 8] and bodies[0]["k"] == 2 and bodies[0]["vector"] == [1.0, 2.0, 3.0, 4.0]
    assert t.vector_search([1, 2, 3, 4]).limit(2).approx_mode("fast").execute()["_rowid"].tolist() == [0, 1]   # local
    assert t.vector_search([1, 2, 3, 4]).limit(2).use_lsm(False).execute()["_rowid"].tolist() == [0, 1]       # local
    assert len(bodies) == 1
    # an evaluated row-id mask has no field on the wire (the wire carries SQL): such a query stays local
    t.vector_search([1, 2, 3, 4]).limit(2).only_if_rowids(allow=[3, 4]).execute()
    assert len(bodies) == 1 and t.index.params.n_filter == 2
    # columns / fast_search / order_by travel in the body (remote/table.rs:762-833) and are honoured on the response
    r = t.vector_search([1, 2, 3, 4]).limit(2).select(["_distance"]).fast_search().order_by([("_distance", False)]).execute()
    b = bodies[1]
    assert b["columns"] == ["_distance"] and b["fast_search"] is True
    assert b["order_by"] == [{"column_name": "_distance", "ascending": False, "nulls_first": False}]
    assert list(r) == ["_distance"] and r["_distance"].tolist() == [0.75, 0.5]
    # the server side reads the same fields back
    req = wire.request_from_json(b)
    assert req.select == ["_distance"] and req.fast_search and req.order_by == [("_distance", False)]
    # the deadline covers the round trip and is handed to the endpoint
    seen = []

    def slow(body, timeout=None):
        seen.append(timeout)
        import time as _t
        _t.sleep(0.05)
        return endpoint(body)

    t.pushdown = slow
    with pytest.raises(lancedb_amd.QueryTimeout):
        t.vector_search([1, 2, 3, 4]).limit(2).execute(QueryExecutionOptions(timeout=0.01))
    assert seen == [0.01]


def test_multi_key_order_by_with_a_descending_primary_key():
    """A descending key must not reverse the order its less significant keys established."""
    t = VectorTable(index=_ArrayIndex(6))
    r = t.vector_search(np.zeros((2, 4))).limit(3).order_by([("query_index", False), ("_distance", True)]).execute()
    assert r["query_index"].tolist() == [1, 1, 1, 0, 0, 0]
    assert r["_distance"][:3].tolist() == sorted(r["_distance"][:3].tolist())
    assert r["_distance"][3:].tolist() == sorted(r["_distance"][3:].tolist())
    r = t.vector_search(np.zeros((2, 4))).limit(3).order_by([("query_index", True), ("_distance", False)]).execute()
    assert r["query_index"].tolist() == [0, 0, 0, 1, 1, 1]
    assert r["_distance"][:3].tolist() == sorted(r["_distance"][:3].tolist(), reverse=True)


def test_default_vector_column_and_supported_types():
    """utils/mod.rs:151-198, :289-298."""
    pa = pytest.importorskip("pyarrow")
    from lancedb_amd.query import default_vector_column, supported_vector_data_type
    vec = lambda n, t=pa.float32(): pa.list_(t, n)  # noqa: E731
    schema = pa.schema([("id", pa.int64()), ("emb", vec(768)), ("meta", pa.struct([("small", vec(8)), ("x", pa.utf8())]))])
    assert default_vector_column(schema, 768) == "emb" and default_vector_column(schema, 8) == "meta.small"
    with pytest.raises(lancedb_amd.InvalidInput, match="No vector column found to match with the query vector dimension: 5"):
        default_vector_column(schema, 5)
    with pytest.raises(lancedb_amd.InvalidInput, match="More than one vector columns found"):
        default_vector_column(schema)
    assert supported_vector_data_type(vec(4)) and supported_vector_data_type(vec(4, pa.float16()))
    assert supported_vector_data_type(vec(4, pa.uint8())) and supported_vector_data_type(pa.list_(vec(4)))
    assert not supported_vector_data_type(vec(4, pa.int32())) and not supported_vector_data_type(pa.utf8())


def test_reference_python_query_tests_on_the_mirror():
    """python/python/tests/test_query.py: test_offset (:251-256), test_vector_query_with_no_limit
    (:853-862), test_invalid_nprobes_sync / test_nprobes_works_sync / test_nprobes_min_max_works_sync
    (:917-940), test_search_empty_table (:1990-2004), test_ensure_vector_query_*empty_list (:2007-2016)."""
    t = VectorTable(index=_ArrayIndex(2))  # the reference fixture has two rows
    assert len(t.vector_search([0, 0, 0, 0]).execute()["_rowid"]) == 2
    assert len(t.vector_search([0, 0, 0, 0]).offset(1).execute()["_rowid"]) == 1
    for bad in (0, None, -3):
        with pytest.raises(ValueError, match="Limit is required for ANN/KNN queries"):
            t.vector_search([0, 0, 0, 0]).limit(bad)
    with pytest.raises(ValueError, match="minimum_nprobes must be greater than 0"):
        t.vector_search([0, 0, 0, 0]).minimum_nprobes(0)
    with pytest.raises(ValueError, match="maximum_nprobes must be greater than or equal to minimum_nprobes"):
        t.vector_search([0, 0, 0, 0]).maximum_nprobes(5)
    with pytest.raises(ValueError, match="minimum_nprobes must be less than or equal to maximum_nprobes"):
        t.vector_search([0, 0, 0, 0]).minimum_nprobes(100)
    t.vector_search([0, 0, 0, 0]).nprobes(30).execute()
    t.vector_search([0, 0, 0, 0]).minimum_nprobes(2).maximum_nprobes(4).execute()
    # an empty table answers with no rows
    empty = VectorTable(index=_ArrayIndex(0))
    r = empty.vector_search([1.0, 2.0, 0, 0]).limit(5).execute()
    assert len(r["_rowid"]) == 0 and len(r["_distance"]) == 0
    for bad in ([], [[]]):
        with pytest.raises(ValueError, match="non-empty"):
            t.vector_search(bad)
    # test_query_timeout (:1846-1873): a zero timeout fails the query with "Query timeout"
    from datetime import timedelta
    from lancedb_amd.query import QueryExecutionOptions
    with pytest.raises(Exception, match="Query timeout"):
        t.vector_search([0.0, 0.0, 0, 0]).execute(QueryExecutionOptions(timeout=timedelta(0)))
    assert len(t.vector_search([0.0, 0.0, 0, 0]).execute(QueryExecutionOptions(timeout=timedelta(seconds=30)))["_rowid"]) == 2
    # test_query_builder_batches (:865-897): to_batches(1) -> two batches of one row, to_batches(2) -> one of two
    q2 = t.vector_search([0, 0, 0, 0]).limit(2)
    assert [len(b["_rowid"]) for b in q2.execute_with_options(QueryExecutionOptions(max_batch_length=1))] == [1, 1]
    assert [len(b["_rowid"]) for b in q2.execute_with_options(QueryExecutionOptions(max_batch_length=2))] == [2]


def test_collected_forms_of_the_python_binding():
    """to_batches / to_arrow / to_list / to_pandas (python/python/lancedb/query.py:986-1117, :1771-1790; the
    schema test python/python/tests/test_query.py:1798-1800: `_distance` is float32)."""
    import datetime
    import pyarrow as pa
    t = VectorTable(index=_ArrayIndex(5000))
    q = t.vector_search([0, 0, 0, 0]).limit(2500)
    rd = q.to_batches()
    assert isinstance(rd, pa.RecordBatchReader)
    assert rd.schema == pa.schema([("_rowid", pa.uint64()), ("_distance", pa.float32())])
    assert [b.num_rows for b in rd] == [1024, 1024, 452]
    assert [b.num_rows for b in q.to_batches(1000)] == [1000, 1000, 500]
    with pytest.raises(lancedb_amd.InvalidInput, match="batch_size"):
        q.to_batches(0)
    tab = q.to_arrow()
    assert tab.num_rows == 2500 and tab.column("_rowid").to_pylist()[:3] == [0, 1, 2]
    assert tab.schema.field("_distance").type == pa.float32()
    rows = t.vector_search([0, 0, 0, 0]).limit(3).to_list()
    assert rows == [{"_rowid": 0, "_distance": 0.0}, {"_rowid": 1, "_distance": 1.0}, {"_rowid": 2, "_distance": 2.0}]
    df = t.vector_search([0, 0, 0, 0]).limit(4).select(["_distance"]).to_pandas()
    assert list(df.columns) == ["_distance"] and df["_distance"].tolist() == [0.0, 1.0, 2.0, 3.0]
    assert t.vector_search([0, 0, 0, 0]).limit(4).to_df().shape == (4, 2)
    # several query vectors add query_index (table/query.rs:334-381)
    mq = t.vector_search([0, 0, 0, 0]).add_query_vector([1, 1, 1, 1]).limit(2).to_arrow()
    assert mq.schema.names == ["_rowid", "_distance", "query_index"] and mq.column("query_index").to_pylist() == [0, 0, 1, 1]
    assert mq.schema.field("query_index").type == pa.int32()
    # the timeout (a timedelta in the binding) reaches the engine as its deadline
    t.vector_search([0, 0, 0, 0]).limit(2).to_arrow(timeout=datetime.timedelta(seconds=1.5))
    assert t.index.params.timeout_ms == 1500
    with pytest.raises(lancedb_amd.QueryTimeout, match="Query timeout"):
        t.vector_search([0, 0, 0, 0]).limit(2).to_list(timeout=datetime.timedelta(0))


def test_builder_conveniences_of_the_python_binding():
    """metric() alias, output_schema() without execution, to_query_object() (python/python/lancedb/query.py:1596-1612,
    :1763-1769, :1790-1822)."""
    import pyarrow as pa
    t = VectorTable(index=_ArrayIndex(10))
    q = t.vector_search([0, 0, 0, 0]).metric("cosine").limit(3)
    assert q.request.distance_type == "cosine"
    with pytest.raises(lancedb_amd.InvalidInput):
        t.vector_search([0, 0, 0, 0]).metric("manhattan")
    assert q.output_schema() == pa.schema([("_rowid", pa.uint64()), ("_distance", pa.float32())])
    assert t.index.params is None  # nothing ran
    assert q.select(["_distance"]).output_schema().names == ["_distance"]
    obj = q.to_query_object()
    assert obj is not q.request and obj.limit == 3 and obj.distance_type == "cosine"
    assert np.array_equal(obj.query_vector[0], q.request.query_vector[0]) and obj.query_vector[0] is not q.request.query_vector[0]
    obj.limit = 99
    assert q.request.limit == 3
