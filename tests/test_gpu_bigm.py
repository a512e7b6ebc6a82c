"""8-bit distance tables larger than the LDS (dim 3072 with the default
dim / 16 = 192 sub-vectors, rust/lancedb/src/index/vector.rs:306-310, could not open).  The
production scan walks such rows in slabs of <= 96 table columns, every row's accumulator starting from
the partial sum of the slabs before (k_scan_skew SLABBED, round 4); the generic scan
(MI355_INDEX_GENERIC_SCAN) keeps the tail of the table in global memory (k_scan_pair SPILL) and sums it
after the LDS part.  Both keep the j-ascending sum, so both are bit-exact against the oracle."""
import numpy as np
import pytest

import lancedb_amd
from lancedb_amd import _abi
from oracle import train

pytestmark = pytest.mark.gpu


def _same(got, exp):
    ids, dist, cnt, st = exp
    assert st == 0
    assert (got.counts == cnt).all() and (got.rowids == ids).all() and (got.distances == dist).all()


@pytest.mark.parametrize("generic", [False, True], ids=["slabs", "generic"])
@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
@pytest.mark.parametrize("shape", [(12000, 384, 12, 192), (6000, 3072, 6, 192), (5000, 480, 5, 240)])
def test_tables_beyond_the_lds(oracle, metric, shape, generic):
    n, dim, nlist, m = shape
    s = train.synthetic_index(n, dim, nlist, m, seed=n + m, empty_parts=1, skew=0.7)
    rng = np.random.default_rng(5)
    raw = rng.normal(size=(n, dim)).astype(np.float32)
    g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw, metric=metric, generic_scan=generic)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw, metric=metric)
    q = rng.normal(size=(7, dim)).astype(np.float32)
    for kw in (dict(k=10, nprobe_min=nlist, nprobe_max=nlist), dict(k=1, nprobe_min=1, nprobe_max=1),
               dict(k=300, nprobe_min=3, nprobe_max=3), dict(k=10, nprobe_min=4, nprobe_max=4, refine_factor=5),
               dict(k=70, nprobe_min=2, nprobe_max=nlist)):
        _same(g.search(q, **kw), o.search(q, **kw))
    st = g.stats()
    assert st["scan_variant"] == (_abi.SCAN_PAIR if generic else _abi.SCAN_SKEW) and st["vectors_scanned"] == o.last_vectors_scanned
    # a batch that fills the chip: one work item per (query, partition) instead of row slices
    qb = rng.normal(size=(600, dim)).astype(np.float32)
    _same(g.search(qb, k=5, nprobe_min=nlist, nprobe_max=nlist), o.search(qb, k=5, nprobe_min=nlist, nprobe_max=nlist))
