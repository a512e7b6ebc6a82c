"""Index population on the GPU (mi355_ivfpq_encode) against the CPU oracle
(oracle/ann_oracle.c orc_ivfpq_encode): partition of every row, PQ codes,
partition offsets and the stable row order are integer outputs -> bit-exact.

The reference hands this stage to lance through
Table.create_index(Index::IvfPq(..)) (rust/lancedb/src/table/create_index.rs:
114-151, :283-303); its own tests only check the index afterwards
(python/python/tests/test_index.py: search results / index stats), which the
end-to-end test below does too: encode on the GPU -> open -> search == oracle
search over the oracle-encoded index.
"""
import numpy as np
import pytest

import lancedb_amd
from lancedb_amd import DeviceArray, _abi
from oracle import train

pytestmark = pytest.mark.gpu


def _trained(n, dim, nlist, m, seed, metric="l2"):
    rng = np.random.default_rng(seed)
    cent = rng.normal(size=(max(nlist, 4), dim)).astype(np.float32) * 2
    x = (cent[rng.integers(0, cent.shape[0], size=n)] + rng.normal(size=(n, dim))).astype(np.float32)
    t = train.train_ivfpq(x[: min(n, 4000)], nlist=nlist, m=m, metric=metric, iters=3)
    return x, t["centroids"], t["codebook"]


def _check(oracle, x, cent, cb, metric):
    exp = oracle.ivfpq_encode(x, cent, cb, _abi.METRIC_NAMES[metric])
    got = lancedb_amd.ivfpq_encode(x, cent, cb, metric=metric, return_assign=True)
    for g, e, name in zip(got, exp, ("part_offsets", "codes", "order", "assign")):
        assert g.shape == e.shape, name
        assert (np.asarray(g).astype(np.uint64) == np.asarray(e).astype(np.uint64)).all(), name
    return got


@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
@pytest.mark.parametrize("shape", [(3000, 32, 16, 8), (5000, 64, 37, 16), (1500, 24, 7, 3), (700, 40, 5, 5),
                                   (2000, 96, 12, 2)])
def test_encode_matches_oracle(oracle, metric, shape):
    n, dim, nlist, m = shape
    x, cent, cb = _trained(n, dim, nlist, m, seed=n + m, metric=metric)
    _check(oracle, x, cent, cb, metric)


def test_encode_c3_subvector_shape_and_chunks(oracle):
    """dim 768 / m 96 (C3's sub-vector shape); 70k rows cross the 65536-row chunk
    of both the assignment pass and the stable-position pass."""
    x, cent, cb = _trained(70000, 768, 64, 96, seed=9)
    po, codes, order, assign = _check(oracle, x, cent, cb, "l2")
    assert int(po[-1]) == 70000
    assert sorted(order.tolist()) == list(range(70000))


def test_encode_ties_nan_and_empty_partitions(oracle):
    rng = np.random.default_rng(2)
    dim, nlist, m = 16, 9, 4
    cent = rng.normal(size=(nlist, dim)).astype(np.float32)
    cent[4] = cent[1]  # duplicate centroid: ties go to the lower partition, 4 stays empty
    cent[7] = 1e30     # never the nearest (distance overflows to +inf)
    cb = rng.normal(size=(m, 256, dim // m)).astype(np.float32)
    cb[:, 200] = cb[:, 17]  # duplicate codebook entry: ties go to the lower code
    x = rng.normal(size=(900, dim)).astype(np.float32)
    x[5] = 0.0
    x[6] = np.nan           # every distance NaN: partition 0, code 0
    x[7, 3] = np.inf
    x[10:20] = x[9]         # identical rows keep their source order
    for metric in ("l2", "cosine", "dot"):
        po, codes, order, assign = _check(oracle, x, cent, cb, metric)
        assert po[5] == po[4]
        assert not (codes == 200).any()
        assert assign[6] == 0


def test_encode_single_row_single_partition_and_empty(oracle):
    rng = np.random.default_rng(4)
    cent = rng.normal(size=(1, 8)).astype(np.float32)
    cb = rng.normal(size=(2, 256, 4)).astype(np.float32)
    _check(oracle, rng.normal(size=(1, 8)).astype(np.float32), cent, cb, "l2")
    po, codes, order = lancedb_amd.ivfpq_encode(np.zeros((0, 8), np.float32), cent, cb)
    assert (po == 0).all() and codes.shape == (0, 2) and order.shape == (0,)


def test_encode_device_resident_io(oracle):
    x, cent, cb = _trained(6000, 64, 20, 16, seed=31)
    exp = oracle.ivfpq_encode(x, cent, cb, _abi.METRIC_L2)
    po, codes, order, assign = lancedb_amd.ivfpq_encode(DeviceArray.from_numpy(x), DeviceArray.from_numpy(cent),
                                                        DeviceArray.from_numpy(cb), return_assign=True)
    assert (po == exp[0]).all()
    assert (codes.numpy() == exp[1]).all()
    assert (order.numpy().astype(np.uint64) == exp[2]).all()
    assert (assign.numpy().astype(np.uint32) == exp[3]).all()


def test_encode_then_search_end_to_end(oracle):
    """GPU-encoded index searched on the GPU == oracle-encoded index searched by the oracle."""
    for metric in ("l2", "cosine", "dot"):
        x, cent, cb = _trained(20000, 64, 32, 16, seed=77, metric=metric)
        ids = np.random.default_rng(1).permutation(20000).astype(np.uint64) * 3 + 1
        po, codes, order = lancedb_amd.ivfpq_encode(x, cent, cb, metric=metric)
        order = order.astype(np.int64)
        g = lancedb_amd.IvfPqIndex(cent, cb, po, codes, ids[order], raw_vectors=x[order], metric=metric)
        epo, ecodes, eorder, _ = oracle.ivfpq_encode(x, cent, cb, _abi.METRIC_NAMES[metric])
        eorder = eorder.astype(np.int64)
        o = oracle.OracleIndex(cent, cb, epo, ecodes, ids[eorder], raw_vectors=x[eorder], metric=metric)
        q = x[:16] + np.float32(0.05)
        for kw in (dict(k=10, nprobe_min=4, nprobe_max=4), dict(k=10, nprobe_min=8, nprobe_max=8, refine_factor=4)):
            got, exp = g.search(q, **kw), o.search(q, **kw)
            assert exp[3] == 0
            assert (got.counts == exp[2]).all() and (got.rowids == exp[0]).all() and (got.distances == exp[1]).all()
        # the query rows are index rows: with refine their own row id comes back first
        got = g.search(x[:16], k=1, nprobe_min=8, nprobe_max=8, refine_factor=10)
        assert (got.rowids[:, 0] == ids[:16]).mean() > 0.9
