"""4-bit PQ end to end (IvfPqIndexBuilder num_bits = 4 with an even num_sub_vectors,
rust/lancedb/src/table/create_index.rs:96-101; python/python/tests/test_index.py:401-420):
open, 16-entry distance tables, two codes per byte in the ADC scan, encode, builder — bit-exact
against the oracle."""
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


@pytest.mark.parametrize("generic", [False, True], ids=["production", "generic"])
@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
@pytest.mark.parametrize("shape", [(20000, 64, 16, 16), (9000, 96, 8, 96), (3000, 24, 5, 2), (12000, 256, 6, 128)])
def test_pq4_search_matches_oracle(oracle, metric, shape, generic):
    """Round 4: 4-bit indexes run the production scan too — the nibbles are expanded to one byte per sub-quantiser when the
    streams are packed and the table has 16 rows (`generic`: the packed-nibble generic kernel, MI355_INDEX_GENERIC_SCAN)."""
    n, dim, nlist, m = shape
    s = train.synthetic_index(n, dim, nlist, m, seed=n + m, empty_parts=1, nbits=4)
    assert s["codes"].shape == (n, m // 2) and s["codebook"].shape == (m, 16, dim // m)
    g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], metric=metric, nbits=4,
                               generic_scan=generic)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], metric=metric, nbits=4)
    q = np.random.default_rng(3).normal(size=(9, dim)).astype(np.float32)
    for nprobe in (1, max(1, nlist // 2), nlist):
        for k in (1, 10, 70, 300):
            _same(g.search(q, k=k, nprobe_min=nprobe, nprobe_max=nprobe), o.search(q, k=k, nprobe_min=nprobe, nprobe_max=nprobe))
    st = g.stats()
    # (m = 6 at 4 bits would stream 32 bytes per 3-byte row: the handle keeps the generic layout for it, ann_index_open.hip)
    production = _abi.SCAN_SKEW if 32 * ((m + 95) // 96) <= 8 * (m // 2) or m > 32 else _abi.SCAN_PAIR
    assert st["scan_variant"] == (_abi.SCAN_PAIR if generic else production) and st["vectors_scanned"] == o.last_vectors_scanned
    assert st["code_bytes_scanned"] == o.last_vectors_scanned * (m // 2)  # algorithmic bytes: m * nbits / 8
    # lance's per-partition transposed storage of the packed bytes
    tr = train.to_part_transposed(s["codes"], s["part_offsets"])
    g2 = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], tr, s["row_ids"], metric=metric, nbits=4,
                                codes_layout=_abi.CODES_PART_TRANSPOSED, generic_scan=generic)
    _same(g2.search(q, k=10, nprobe_min=nlist, nprobe_max=nlist), o.search(q, k=10, nprobe_min=nlist, nprobe_max=nlist))


def test_pq4_rejects_odd_sub_vectors():
    s = train.synthetic_index(100, 9, 2, 3, seed=1)
    with pytest.raises(lancedb_amd.InvalidInput, match="even when num_bits is 4"):
        lancedb_amd.IvfPqIndex(s["centroids"], np.zeros((3, 16, 3), np.float32), s["part_offsets"],
                               np.zeros((100, 1), np.uint8), nbits=4)


@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
def test_pq4_encode_matches_oracle_and_round_trips(oracle, metric):
    rng = np.random.default_rng(11)
    n, dim, nlist, m = 5000, 48, 12, 12
    x = rng.normal(size=(n, dim)).astype(np.float32)
    cen = rng.normal(size=(nlist, dim)).astype(np.float32)
    cb = rng.normal(0, 0.6, size=(m, 16, dim // m)).astype(np.float32)
    po, codes, order, assign = lancedb_amd.ivfpq_encode(x, cen, cb, metric=metric, return_assign=True, nbits=4)
    epo, ecodes, eorder, eassign = oracle.ivfpq_encode(x, cen, cb, metric=metric, nbits=4)
    assert codes.shape == (n, m // 2)
    assert (po == epo).all() and (order == eorder).all() and (assign == eassign).all() and (codes == ecodes).all()
    g = lancedb_amd.IvfPqIndex(cen, cb, po, codes, order, raw_vectors=x[order.astype(np.int64)], metric=metric, nbits=4)
    o = oracle.OracleIndex(cen, cb, po, codes, order, raw_vectors=x[order.astype(np.int64)], metric=metric, nbits=4)
    q = rng.normal(size=(6, dim)).astype(np.float32)
    for kw in (dict(k=10, nprobe_min=4, nprobe_max=4), dict(k=5, nprobe_min=12, nprobe_max=12, refine_factor=8)):
        _same(g.search(q, **kw), o.search(q, **kw))


def test_pq4_builder(oracle):
    """IvfPqBuilder(num_bits=4): trains 16-entry codebooks, encodes, opens; recall sanity on clustered data."""
    rng = np.random.default_rng(3)
    cent = rng.normal(size=(32, 32)).astype(np.float32) * 4
    x = (cent[rng.integers(0, 32, size=20000)] + rng.normal(size=(20000, 32))).astype(np.float32)
    b = lancedb_amd.IvfPqBuilder(num_partitions=16, num_sub_vectors=16, num_bits=4, max_iterations=6)
    ix = b.build(x)
    assert ix.nbits == 4
    q = x[rng.integers(0, 20000, size=40)]
    got = ix.search(q, k=10, nprobe_min=16, nprobe_max=16, refine_factor=20)
    truth, _, _, _ = oracle.flat_search(x, q, k=10)
    recall = np.mean([len(set(truth[i]) & set(got.rowids[i])) / 10 for i in range(40)])
    assert recall > 0.8
