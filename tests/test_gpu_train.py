"""Index training on the GPU (mi355_kmeans_train, mi355_ivf_residuals, the
IvfPqBuilder host mirror) against the CPU oracle.  The Lloyd iteration is
deterministic by definition (include/mi355_ann.h): trained centroids are
compared with == on their f32 bit patterns, counts and assignments as integers.

The reference exposes training only through Table.create_index and checks the
result through searches / index statistics (python/python/tests/test_index.py,
rust/lancedb/src/index/vector.rs:61-119 for the parameters); the end-to-end
tests below do the same on top of the bit-exact stage checks.
"""
import numpy as np
import pytest

import lancedb_amd
from lancedb_amd import DeviceArray, _abi

pytestmark = pytest.mark.gpu


def _clustered(n, dim, nc, seed, spread=1.0):
    rng = np.random.default_rng(seed)
    cent = rng.normal(size=(nc, dim)).astype(np.float32) * 3
    x = (cent[rng.integers(0, nc, size=n)] + rng.normal(size=(n, dim)).astype(np.float32) * np.float32(spread))
    return np.ascontiguousarray(x, dtype=np.float32)


def _same_bits(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    nan = np.isnan(a)  # NaN payloads are not part of the contract; their positions are
    return (nan == np.isnan(b)).all() and (a.view(np.uint32)[~nan] == b.view(np.uint32)[~nan]).all()


@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
@pytest.mark.parametrize("shape", [(5000, 32, 16), (3000, 24, 7), (9000, 64, 100), (70000, 16, 33)])
def test_kmeans_matches_oracle(oracle, metric, shape):
    n, dim, k = shape
    x = _clustered(n, dim, max(k, 8), seed=n + k)
    init = x[np.random.default_rng(1).choice(n, size=k, replace=False)]
    for iters in (1, 4):
        exp_c, exp_n = oracle.kmeans_train(x, init, metric, iters)
        got_c, got_n = lancedb_amd.kmeans_train(x, init, metric=metric, iters=iters)
        assert (got_n == exp_n).all()
        assert _same_bits(got_c, exp_c)
        assert int(got_n.sum()) == n


def test_kmeans_pq_subquantiser_on_strided_columns(oracle):
    """k = 256 on columns [16, 24) of a 64-wide residual matrix (ld = 64)."""
    x = _clustered(20000, 64, 40, seed=3, spread=0.7)
    init = np.ascontiguousarray(x[np.random.default_rng(5).choice(20000, size=256, replace=False), 16:24])
    exp_c, exp_n = oracle.kmeans_train(x, init, "l2", 5, cols=(16, 24))
    got_c, got_n = lancedb_amd.kmeans_train(x, init, metric="l2", iters=5, cols=(16, 24))
    assert (got_n == exp_n).all() and _same_bits(got_c, exp_c)
    # same as training on a dense copy of the columns
    dense_c, _ = lancedb_amd.kmeans_train(np.ascontiguousarray(x[:, 16:24]), init, iters=5)
    assert _same_bits(dense_c, got_c)


def test_kmeans_empty_centroid_ties_few_rows_and_zero_iters(oracle):
    rng = np.random.default_rng(8)
    x = rng.normal(size=(500, 8)).astype(np.float32)
    init = x[:6].copy()
    init[3] = 1e6          # never the nearest: keeps its value, count 0
    init[5] = init[1]      # duplicate: ties go to the lower centroid, 5 stays empty
    exp_c, exp_n = oracle.kmeans_train(x, init, "l2", 3)
    got_c, got_n = lancedb_amd.kmeans_train(x, init, iters=3)
    assert (got_n == exp_n).all() and _same_bits(got_c, exp_c)
    assert got_n[3] == 0 and _same_bits(got_c[3], init[3])
    one_c, one_n = lancedb_amd.kmeans_train(x, init, iters=1)  # the duplicate loses every tie in the first pass
    assert one_n[5] == 0 and _same_bits(one_c[5], init[5]) and one_n[1] > 0
    # fewer rows than centroids
    exp_c, exp_n = oracle.kmeans_train(x[:4], init, "l2", 2)
    got_c, got_n = lancedb_amd.kmeans_train(x[:4], init, iters=2)
    assert (got_n == exp_n).all() and _same_bits(got_c, exp_c)
    # zero iterations / zero rows: centroids untouched, counts zero
    for rows in (x, x[:0]):
        got_c, got_n = lancedb_amd.kmeans_train(rows, init, iters=0 if rows.shape[0] else 3)
        assert _same_bits(got_c, init) and (got_n == 0).all()
    # a NaN row is assigned to centroid 0 and poisons only that centroid
    x2 = x.copy()
    x2[17] = np.nan
    exp_c, exp_n = oracle.kmeans_train(x2, init, "l2", 2)
    got_c, got_n = lancedb_amd.kmeans_train(x2, init, iters=2)
    assert (got_n == exp_n).all() and _same_bits(got_c, exp_c)


def test_kmeans_objective_never_increases():
    x = _clustered(8000, 16, 12, seed=4)
    init = x[np.random.default_rng(2).choice(8000, size=12, replace=False)]
    prev = np.inf
    for iters in range(0, 6):
        c, _ = lancedb_amd.kmeans_train(x, init, iters=iters)
        d2 = ((x[:, None, :].astype(np.float64) - c[None].astype(np.float64)) ** 2).sum(-1).min(1).sum()
        assert d2 <= prev * (1 + 1e-6)
        prev = d2


@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
def test_residuals_match_oracle(oracle, metric):
    x = _clustered(7000, 48, 20, seed=6)
    cent = x[np.random.default_rng(3).choice(7000, size=20, replace=False)]
    exp_r, exp_a = oracle.ivf_residuals(x, cent, metric)
    got_r, got_a = lancedb_amd.ivf_residuals(x, cent, metric=metric)
    assert (got_a == exp_a).all() and _same_bits(got_r, exp_r)


def test_training_device_resident_io(oracle):
    x = _clustered(6000, 32, 10, seed=12)
    init = x[:10].copy()
    exp_c, exp_n = oracle.kmeans_train(x, init, "l2", 3)
    dx = DeviceArray.from_numpy(x)
    got_c, got_n = lancedb_amd.kmeans_train(dx, DeviceArray.from_numpy(init), iters=3)
    assert _same_bits(got_c.numpy(), exp_c) and (got_n.numpy().astype(np.uint64) == exp_n).all()
    exp_r, exp_a = oracle.ivf_residuals(x, exp_c, "l2")
    got_r, got_a = lancedb_amd.ivf_residuals(dx, got_c)
    assert _same_bits(got_r.numpy(), exp_r) and (got_a.numpy().astype(np.uint32) == exp_a).all()


def test_builder_end_to_end_recall_and_oracle_parity(oracle):
    """create_index-shaped build on the GPU, searched on the GPU; the same trained arrays
    opened by the oracle give identical results, and the index is a useful one."""
    x = _clustered(40000, 64, 64, seed=21)
    q = x[np.random.default_rng(9).choice(40000, size=64, replace=False)] + np.float32(0.01)
    for metric in ("l2", "cosine"):
        b = lancedb_amd.IvfPqBuilder(distance_type=metric, num_partitions=64, num_sub_vectors=16, max_iterations=8,
                                     sample_rate=64)
        cent, cb = b.train(x)
        assert cent.shape == (64, 64) and cb.shape == (16, 256, 4)
        po, codes, order = lancedb_amd.ivfpq_encode(x, cent, cb, metric=metric)
        order = order.astype(np.int64)
        g = lancedb_amd.IvfPqIndex(cent, cb, po, codes, order.astype(np.uint64), raw_vectors=x[order], metric=metric)
        o = oracle.OracleIndex(cent, cb, po, codes, order.astype(np.uint64), raw_vectors=x[order], metric=metric)
        kw = dict(k=10, nprobe_min=8, nprobe_max=8, refine_factor=10)
        got, exp = g.search(q, **kw), o.search(q, **kw)
        assert (got.rowids == exp[0]).all() and (got.distances == exp[1]).all()
        truth = oracle.flat_search(x, q, k=10, metric=_abi.METRIC_NAMES[metric])[0]
        recall = np.mean([len(set(got.rowids[i].tolist()) & set(truth[i].tolist())) / 10 for i in range(len(q))])
        assert recall > 0.9, recall
        # partitions are balanced enough to be useful: no partition holds more than 10x its share
        lens = np.diff(po.astype(np.int64))
        assert lens.max() < 10 * 40000 / 64
    # the one-call form
    ix = lancedb_amd.IvfPqBuilder(num_partitions=32, num_sub_vectors=8, max_iterations=4, sample_rate=32).build(x)
    got = ix.search(x[:8], k=1, nprobe_min=8, nprobe_max=8, refine_factor=10)
    assert (got.rowids[:, 0] == np.arange(8)).all()
