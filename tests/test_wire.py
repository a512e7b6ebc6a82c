"""Remote wire format (rust/lancedb/src/remote/table.rs:724-930): the JSON body the
reference's client sends and the Arrow IPC file it expects back."""
import json

import numpy as np
import pytest

from lancedb_amd import wire
from lancedb_amd.query import VectorQueryRequest


def test_default_body_matches_the_reference_pin():
    """remote/table.rs:4640-4699 test_query_vector_default_values."""
    req = VectorQueryRequest(limit=10, query_vector=[np.array([0.1, 0.2, 0.3], np.float32)])
    expected = {"prefilter": True, "nprobes": 20, "minimum_nprobes": 20, "maximum_nprobes": 20,
                "lower_bound": None, "upper_bound": None, "k": 10, "ef": None, "refine_factor": None,
                "version": None, "vector": [float(np.float32(x)) for x in (0.1, 0.2, 0.3)]}
    assert wire.request_to_json(req) == expected
    back = wire.request_from_json(json.dumps(expected))
    assert back.limit == 10 and back.minimum_nprobes == 20 and back.maximum_nprobes == 20
    assert back.prefilter and back.use_index and back.refine_factor is None
    assert (back.query_vector[0] == np.array([0.1, 0.2, 0.3], np.float32)).all()


def test_approx_mode_is_sent_when_set():
    """remote/table.rs:4694-4745 test_query_vector_approx_mode_sent_when_set."""
    req = VectorQueryRequest(limit=10, query_vector=[np.array([0.1, 0.2, 0.3], np.float32)], approx_mode="accurate")
    expected = {"prefilter": True, "nprobes": 20, "minimum_nprobes": 20, "maximum_nprobes": 20, "approx_mode": "accurate",
                "lower_bound": None, "upper_bound": None, "k": 10, "ef": None, "refine_factor": None,
                "version": None, "vector": [float(np.float32(x)) for x in (0.1, 0.2, 0.3)]}
    assert wire.request_to_json(req) == expected
    assert wire.request_from_json(expected).approx_mode == "accurate"
    with pytest.raises(Exception, match="approx_mode must be one of"):
        wire.request_from_json({**expected, "approx_mode": "sloppy"})


def test_round_trip_of_every_field():
    req = VectorQueryRequest(limit=7, offset=2, with_row_id=True, column="emb",
                             query_vector=[np.arange(4, dtype=np.float32), np.ones(4, np.float32)],
                             minimum_nprobes=5, maximum_nprobes=None, lower_bound=0.5, upper_bound=2.0, ef=None,
                             refine_factor=3, distance_type="cosine", use_index=False, prefilter=False)
    body = wire.request_to_json(req)
    assert body["maximum_nprobes"] == 0 and body["bypass_vector_index"] is True and body["vector_column"] == "emb"
    assert isinstance(body["vector"][0], list)  # multivector form
    back = wire.request_from_json(body)
    for f in ("limit", "offset", "with_row_id", "column", "minimum_nprobes", "maximum_nprobes", "lower_bound",
              "upper_bound", "refine_factor", "distance_type", "use_index", "prefilter"):
        assert getattr(back, f) == getattr(req, f), f
    assert len(back.query_vector) == 2
    # an old client only sends `nprobes` (remote/table.rs:846-851)
    old = wire.request_from_json({"k": 3, "nprobes": 9, "vector": [0.0, 1.0]})
    assert (old.minimum_nprobes, old.maximum_nprobes) == (9, 9)


def test_filter_needs_evaluated_row_ids():
    from lancedb_amd import NotSupported
    with pytest.raises(NotSupported):
        wire.request_from_json({"k": 3, "vector": [0.0], "filter": "id % 2 == 0"})


def test_ipc_response_schema():
    pa = pytest.importorskip("pyarrow")
    cols = {"_rowid": np.array([5, 9], np.uint64), "_distance": np.array([0.5, 1.5], np.float32),
            "query_index": np.array([0, 1], np.int32)}
    data = wire.response_to_ipc(cols)
    t = pa.ipc.open_file(pa.BufferReader(data)).read_all()
    assert t.schema.names == ["_rowid", "_distance", "query_index"]
    assert t.schema.field("_rowid").type == pa.uint64() and t.schema.field("_distance").type == pa.float32()
    assert t.column("_rowid").to_pylist() == [5, 9]


@pytest.mark.gpu
def test_handle_query_end_to_end(oracle):
    pa = pytest.importorskip("pyarrow")
    import lancedb_amd
    from oracle import train
    s = train.synthetic_index(20000, 64, 16, 32, seed=2)
    g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    t = lancedb_amd.VectorTable(index=g)
    q = np.random.default_rng(0).normal(size=64).astype(np.float32)
    req = VectorQueryRequest(limit=5, offset=1, query_vector=[q], minimum_nprobes=4, maximum_nprobes=4)
    even = s["row_ids"][s["row_ids"] % 2 == 0]
    body = dict(wire.request_to_json(req), filter="id % 2 == 0")
    ctype, data = wire.handle_query(t, json.dumps(body), allow_rowids=even)
    assert ctype == wire.ARROW_FILE_CONTENT_TYPE
    tab = pa.ipc.open_file(pa.BufferReader(data)).read_all()
    exp = o.search(q[None], k=6, nprobe_min=4, nprobe_max=4, allow_rowids=even)
    assert tab.column("_rowid").to_pylist() == exp[0][0, 1:6].tolist()
    assert np.array_equal(np.array(tab.column("_distance").to_pylist(), np.float32), exp[1][0, 1:6])
