"""Builder semantics of the VectorQuery mirror, written after the reference's
own unit tests (rust/lancedb/src/query.rs:1232-1288, nodejs/__test__/table.test.ts:967-972)."""
import numpy as np
import pytest

import lancedb_amd
from lancedb_amd import VectorQuery, VectorQueryRequest
from lancedb_amd.query import VectorTable


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
