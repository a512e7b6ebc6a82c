"""Parity of the HIP path (through the C ABI) against the CPU oracle.

Bar (BASELINE.json north_star): returned row ids bit-exact, distances within
1e-4 relative.  The kernels follow the oracle's arithmetic contract, so the
distances are in fact compared for exact equality wherever the chain order is
fixed by the contract.
"""
import numpy as np
import pytest

import lancedb_amd
from lancedb_amd import _abi
from oracle import train

pytestmark = pytest.mark.gpu

RTOL = 1e-4  # north_star tolerance on distances


def _both(oracle, s, metric="l2", raw=None, layout=_abi.CODES_ROW_MAJOR, codes=None):
    codes = s["codes"] if codes is None else codes
    g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], codes, s.get("row_ids"),
                               raw_vectors=raw, metric=metric, codes_layout=layout)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], codes, s.get("row_ids"),
                           raw_vectors=raw, metric=metric, codes_layout=layout)
    return g, o


def _assert_same(got, exp, exact_dist=True):
    ids, dist, cnt, st = exp
    assert st == 0
    assert (got.counts == cnt).all()
    assert (got.rowids == ids).all()
    if exact_dist:
        assert (got.distances == dist).all()
    else:
        fin = np.isfinite(dist)
        np.testing.assert_allclose(got.distances[fin], dist[fin], rtol=RTOL, atol=0)
        assert (got.distances[~fin] == dist[~fin]).all()


@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
@pytest.mark.parametrize("shape", [(5000, 32, 16, 8), (20000, 64, 64, 16), (3000, 24, 7, 3)])
def test_ivfpq_matches_oracle(oracle, metric, shape):
    n, dim, nlist, m = shape
    s = train.synthetic_index(n, dim, nlist, m, seed=n + m, empty_parts=min(2, nlist // 4))
    g, o = _both(oracle, s, metric)
    q = np.random.default_rng(3).normal(size=(9, dim)).astype(np.float32)
    for nprobe in (1, max(1, nlist // 4), nlist):
        for k in (1, 10, 70):
            _assert_same(g.search(q, k=k, nprobe_min=nprobe, nprobe_max=nprobe),
                         o.search(q, k=k, nprobe_min=nprobe, nprobe_max=nprobe))


def test_ivfpq_c3_shaped_partition_sizes(oracle):
    """dim 768 / m 96 (the C3 sub-vector shape) with partitions long enough to
    exercise the 16-rows-per-thread path, several slices and ragged tails."""
    s = train.synthetic_index(120000, 768, 8, 96, seed=5, skew=0.8)
    g, o = _both(oracle, s)
    q = (s["centroids"][:6] + np.random.default_rng(1).normal(0, 0.5, size=(6, 768))).astype(np.float32)
    _assert_same(g.search(q, k=10, nprobe_min=4, nprobe_max=4), o.search(q, k=10, nprobe_min=4, nprobe_max=4))
    g.configure(slice_rows=4096)
    _assert_same(g.search(q, k=100, nprobe_min=8, nprobe_max=8), o.search(q, k=100, nprobe_min=8, nprobe_max=8))
    st = g.stats()
    assert st["vectors_scanned"] == o.last_vectors_scanned
    assert st["code_bytes_scanned"] == o.last_vectors_scanned * 96


def test_ivfpq_trained_index_with_refine_and_recall(oracle):
    rng = np.random.default_rng(21)
    cent = rng.normal(size=(64, 64)).astype(np.float32) * 3
    x = (cent[rng.integers(0, 64, size=30000)] + rng.normal(size=(30000, 64))).astype(np.float32)
    for metric in ("l2", "cosine"):
        t = train.train_ivfpq(x, nlist=32, m=16, metric=metric, iters=5)
        g, o = _both(oracle, t, metric, raw=t["raw"])
        q = (cent[rng.integers(0, 64, size=32)] + rng.normal(size=(32, 64))).astype(np.float32)
        _assert_same(g.search(q, k=10, nprobe_min=8, nprobe_max=8), o.search(q, k=10, nprobe_min=8, nprobe_max=8))
        got = g.search(q, k=10, nprobe_min=8, nprobe_max=8, refine_factor=5)
        _assert_same(got, o.search(q, k=10, nprobe_min=8, nprobe_max=8, refine_factor=5))
        truth, _, _, _ = oracle.flat_search(x, q, k=10, metric=_abi.METRIC_NAMES[metric])
        recall = np.mean([len(set(truth[i]) & set(got.rowids[i])) / 10 for i in range(32)])
        assert recall > 0.7  # sanity only: the index is trained for 5 Lloyd iterations


def test_ivfpq_ties_duplicates_and_ranges(oracle):
    """Collisions: every row of a partition has the same code -> identical
    distances; the order must come from the row id alone."""
    s = train.synthetic_index(6000, 32, 8, 8, seed=2)
    s["codes"][:] = s["codes"][0]
    s["row_ids"] = np.random.default_rng(0).permutation(6000).astype(np.uint64) + (1 << 40)
    g, o = _both(oracle, s)
    q = np.random.default_rng(4).normal(size=(4, 32)).astype(np.float32)
    _assert_same(g.search(q, k=25, nprobe_min=3, nprobe_max=3), o.search(q, k=25, nprobe_min=3, nprobe_max=3))
    # distance_range [lower, upper) on a normal index
    s2 = train.synthetic_index(8000, 32, 8, 8, seed=9)
    g2, o2 = _both(oracle, s2)
    ids, dist, cnt, _ = o2.search(q, k=20, nprobe_min=8, nprobe_max=8)
    lo, hi = float(dist[0, 3]), float(dist[0, 12])
    kw = dict(k=20, nprobe_min=8, nprobe_max=8, lower_bound=lo, upper_bound=hi)
    got = g2.search(q, **kw)
    _assert_same(got, o2.search(q, **kw))
    assert got.counts[0] == 9 and got.distances[0, 0] == lo and (got.distances[0, :9] < hi).all()
    # NaN query -> NULL distances -> no rows
    qn = q.copy()
    qn[1, 5] = np.nan
    _assert_same(g2.search(qn, k=5, nprobe_min=2, nprobe_max=2), o2.search(qn, k=5, nprobe_min=2, nprobe_max=2))


def test_ivfpq_small_and_ragged(oracle):
    # k larger than the index, empty partitions, nprobe > nlist, identity row ids
    s = train.synthetic_index(37, 8, 4, 2, seed=1, empty_parts=1)
    del s["row_ids"]
    g, o = _both(oracle, s)
    q = np.zeros((2, 8), np.float32)
    _assert_same(g.search(q, k=64, nprobe_min=9, nprobe_max=9), o.search(q, k=64, nprobe_min=9, nprobe_max=9))
    # maximum_nprobes expansion only for queries that come back short
    _assert_same(g.search(q, k=30, nprobe_min=1, nprobe_max=4), o.search(q, k=30, nprobe_min=1, nprobe_max=4))
    _assert_same(g.search(q, k=30, nprobe_min=1, nprobe_max=None), o.search(q, k=30, nprobe_min=1, nprobe_max=None))
    # k = 0 and zero queries
    r = g.search(q, k=0, nprobe_min=1, nprobe_max=1)
    assert (r.counts == 0).all()
    # lance's transposed code layout gives identical results
    t = train.to_part_transposed(s["codes"], s["part_offsets"])
    g2, o2 = _both(oracle, s, layout=_abi.CODES_PART_TRANSPOSED, codes=t)
    _assert_same(g2.search(q, k=10, nprobe_min=4, nprobe_max=4), o.search(q, k=10, nprobe_min=4, nprobe_max=4))


# ------------------------------------------------- skewed (production) scan --
@pytest.mark.parametrize("m", [32, 48, 64, 80, 96])
@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
def test_ivfpq_skew_layout_matches_oracle(oracle, m, metric):
    """m in {32,48,64,80,96} takes the pre-skewed stream layout + conflict-free
    table (kernels_skew.h).  Partition lengths cover: empty, < 1 tile, exactly
    16 tiles, ragged tails, > 16 tiles per stream."""
    dim, nlist = m * 4, 12
    rng = np.random.default_rng(m)
    lens = np.array([0, 1, 63, 64, 65, 1024, 1025, 3000, 0, 5000, 17, 2047], dtype=np.int64)
    n = int(lens.sum())
    s = train.synthetic_index(n, dim, nlist, m, seed=m)
    s["part_offsets"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    g, o = _both(oracle, s, metric)
    assert g.stats()["scan_variant"] in (0, _abi.SCAN_SKEW)
    q = (s["centroids"][rng.integers(0, nlist, size=7)] + rng.normal(0, 0.5, size=(7, dim))).astype(np.float32)
    for nprobe, k in ((1, 10), (5, 1), (12, 10), (12, 64), (12, 100), (12, 200)):
        _assert_same(g.search(q, k=k, nprobe_min=nprobe, nprobe_max=nprobe),
                     o.search(q, k=k, nprobe_min=nprobe, nprobe_max=nprobe))
    assert g.stats()["scan_variant"] == _abi.SCAN_SKEW
    # lance's per-partition transposed source layout packs to the same streams
    t = train.to_part_transposed(s["codes"], s["part_offsets"])
    g2, _ = _both(oracle, s, metric, layout=_abi.CODES_PART_TRANSPOSED, codes=t)
    _assert_same(g2.search(q, k=10, nprobe_min=12, nprobe_max=12), o.search(q, k=10, nprobe_min=12, nprobe_max=12))


def test_ivfpq_skew_ties_ranges_and_batches(oracle):
    """All rows of a partition share one code (every distance ties: order comes
    from the row id alone), distance ranges, and a batch large enough that every
    queue of the partition-major work list is used and stolen from."""
    s = train.synthetic_index(30000, 128, 24, 32, seed=77, skew=1.0)
    s["codes"][:] = s["codes"][0]
    s["row_ids"] = np.random.default_rng(0).permutation(30000).astype(np.uint64) + (1 << 40)
    g, o = _both(oracle, s)
    q = np.random.default_rng(4).normal(size=(300, 128)).astype(np.float32)
    _assert_same(g.search(q, k=25, nprobe_min=6, nprobe_max=6), o.search(q, k=25, nprobe_min=6, nprobe_max=6))
    s2 = train.synthetic_index(50000, 192, 32, 48, seed=9, skew=0.7)
    g2, o2 = _both(oracle, s2)
    q2 = (s2["centroids"][np.random.default_rng(1).integers(0, 32, size=300)]
          + np.random.default_rng(2).normal(0, 0.5, size=(300, 192))).astype(np.float32)
    exp = o2.search(q2, k=10, nprobe_min=8, nprobe_max=8)
    _assert_same(g2.search(q2, k=10, nprobe_min=8, nprobe_max=8), exp)
    lo, hi = float(exp[1][0, 2]), float(exp[1][0, 8])
    kw = dict(k=10, nprobe_min=8, nprobe_max=8, lower_bound=lo, upper_bound=hi)
    _assert_same(g2.search(q2, **kw), o2.search(q2, **kw))
    qn = q2[:4].copy()
    qn[1, 5] = np.nan
    _assert_same(g2.search(qn, k=5, nprobe_min=2, nprobe_max=2), o2.search(qn, k=5, nprobe_min=2, nprobe_max=2))
    # maximum_nprobes expansion and refine on the skewed layout
    raw = np.random.default_rng(3).normal(size=(50000, 192)).astype(np.float32)
    g3, o3 = _both(oracle, s2, raw=raw)
    _assert_same(g3.search(q2[:16], k=10, nprobe_min=4, nprobe_max=4, refine_factor=10),
                 o3.search(q2[:16], k=10, nprobe_min=4, nprobe_max=4, refine_factor=10))


def test_generic_layout_still_serves_m96(oracle):
    """MI355_INDEX_GENERIC_SCAN keeps the generic [m][rows] layout for a supported m."""
    s = train.synthetic_index(20000, 192, 8, 96, seed=5)
    g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], generic_scan=True)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    q = np.random.default_rng(1).normal(size=(5, 192)).astype(np.float32)
    _assert_same(g.search(q, k=10, nprobe_min=4, nprobe_max=4), o.search(q, k=10, nprobe_min=4, nprobe_max=4))
    assert g.stats()["scan_variant"] == _abi.SCAN_PAIR


def test_prefilter_allow_and_block_lists(oracle):
    """Prefilter (the reference's default for filtered vector queries,
    rust/lancedb/src/query.rs:489-507; row counts pinned at :1759-1812): the k nearest
    among the permitted rows, on both scan kernels, with refine, with maximum_nprobes
    expansion, on the flat path and through the query mirror (pre- and post-filter)."""
    rng = np.random.default_rng(31)
    for m, dim in ((8, 32), (48, 192)):  # generic and skewed layouts
        s = train.synthetic_index(40000, dim, 32, m, seed=m, skew=0.6)
        raw = rng.normal(size=(40000, dim)).astype(np.float32)
        g, o = _both(oracle, s, raw=raw)
        q = (s["centroids"][rng.integers(0, 32, size=40)] + rng.normal(0, 0.5, size=(40, dim))).astype(np.float32)
        ids = s["row_ids"]
        allow_few = rng.choice(ids, size=37, replace=False)        # fewer permitted rows than k in most partitions
        allow_many = rng.choice(ids, size=30000, replace=False)
        block = rng.choice(ids, size=39000, replace=False)
        for kw in (dict(allow_rowids=allow_many), dict(allow_rowids=allow_few), dict(block_rowids=block)):
            for extra in (dict(nprobe_min=8, nprobe_max=8), dict(nprobe_min=2, nprobe_max=32),
                          dict(nprobe_min=8, nprobe_max=8, refine_factor=4)):
                got = g.search(q, k=10, **extra, **kw)
                exp = o.search(q, k=10, **extra, **kw)
                _assert_same(got, exp)
                flt = np.asarray(list(kw.values())[0], dtype=np.uint64)
                valid = got.rowids[got.rowids != _abi.UINT64_MAX]
                assert np.isin(valid, flt).all() if "allow_rowids" in kw else not np.isin(valid, flt).any()
        # flat path
        f = lancedb_amd.FlatIndex(raw)
        _assert_same(f.search(q, k=10, allow_rowids=np.arange(0, 40000, 7)),
                     oracle.flat_search(raw, q, k=10, allow_rowids=np.arange(0, 40000, 7)))
        assert f.info()[0] == 2  # prefiltered flat searches take the exact sweep
        # mirror: prefilter returns 10 permitted rows, postfilter thins the unfiltered top 10
        t = lancedb_amd.VectorTable(index=g, flat=f)
        pre = t.vector_search(q[0]).nprobes(8).limit(10).only_if_rowids(block=block).execute()
        assert len(pre["_rowid"]) == 10 and not np.isin(pre["_rowid"], block).any()
        post = t.vector_search(q[0]).nprobes(8).limit(10).only_if_rowids(block=block).postfilter().execute()
        plain = t.vector_search(q[0]).nprobes(8).limit(10).execute()
        assert post["_rowid"].tolist() == [r for r in plain["_rowid"].tolist() if r not in set(block.tolist())]


def test_ivfpq_errors_mirror_reference(oracle):
    s = train.synthetic_index(500, 8, 4, 2, seed=1)
    g, _ = _both(oracle, s)
    q = np.zeros((1, 8), np.float32)
    with pytest.raises(lancedb_amd.InvalidInput, match="minimum_nprobes must be greater than 0"):
        g.search(q, k=5, nprobe_min=0, nprobe_max=4)
    with pytest.raises(lancedb_amd.InvalidInput, match="maximum_nprobes must be greater than or equal"):
        g.search(q, k=5, nprobe_min=5, nprobe_max=4)
    with pytest.raises(lancedb_amd.InvalidInput, match="distance type"):
        g.search(q, k=5, metric=_abi.METRIC_COSINE)
    with pytest.raises(lancedb_amd.InvalidInput, match="refine_factor"):
        g.search(q, k=5, refine_factor=2)
    with pytest.raises(lancedb_amd.InvalidInput, match="approx_mode"):
        g.search(q, k=5, approx_mode=9)
    for mode in ("fast", "normal", "accurate"):  # carried, validated, ignored by IVF-PQ (lib.rs:298-313)
        assert (g.search(q, k=5, nprobe_min=4, nprobe_max=4, approx_mode=mode).rowids ==
                g.search(q, k=5, nprobe_min=4, nprobe_max=4).rowids).all()


def test_sharded_handles_merge_to_unsharded_result(oracle):
    """Partition sharding: every shard scans the probed partitions
    it owns; the k-way merge of the per-shard candidates equals the unsharded
    result.  Device buffers come from lancedb_amd.DeviceArray (no torch)."""
    DA = lancedb_amd.DeviceArray
    for m, dim in ((8, 32), (32, 128)):  # generic and skewed layouts
        s = train.synthetic_index(40000, dim, 64, m, seed=13, skew=0.9)
        q = np.random.default_rng(6).normal(size=(33, dim)).astype(np.float32)
        _, o = _both(oracle, s)
        exp = o.search(q, k=10, nprobe_min=16, nprobe_max=16)
        for shards in (2, 8):
            parts = []
            rows = 0
            for r in range(shards):
                g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"],
                                           s["row_ids"], shard_count=shards, shard_rank=r)
                rows += g.info()[0]
                parts.append(g.search(q, k=10, nprobe_min=16, nprobe_max=16))
            assert rows == 40000
            ids = DA.from_numpy(np.stack([p.rowids.astype(np.int64) for p in parts]))
            dist = DA.from_numpy(np.stack([p.distances for p in parts]))
            cnt = DA.from_numpy(np.stack([p.counts.astype(np.int32) for p in parts]))
            mi, md, mc = lancedb_amd.merge_topk(ids, dist, cnt, 10)
            assert (mi.numpy().astype(np.uint64) == exp[0]).all()
            assert (md.numpy() == exp[1]).all()
            assert (mc.numpy().astype(np.uint32) == exp[2]).all()


def test_two_phase_search_with_sharded_coarse_stage(oracle):
    """every shard scores a slice of the centroids; the merged
    (distance, partition id) lists are the probe list of the unsharded search, and the
    merged per-shard scans of those probes equal the unsharded result."""
    from lancedb_amd.distributed import coarse_slice
    DA = lancedb_amd.DeviceArray
    for m, dim, metric in ((8, 32, "l2"), (32, 128, "cosine"), (48, 96, "dot")):
        s = train.synthetic_index(50000, dim, 96, m, seed=17, skew=0.8, empty_parts=5)
        q = (s["centroids"][np.random.default_rng(6).integers(0, 96, size=41)]
             + np.random.default_rng(7).normal(0, 0.5, size=(41, dim))).astype(np.float32)
        _, o = _both(oracle, s, metric)
        for shards, nprobe in ((2, 16), (5, 7), (8, 40)):
            exp = o.search(q, k=10, nprobe_min=nprobe, nprobe_max=nprobe)
            hs = [lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"],
                                         metric=metric, shard_count=shards, shard_rank=r) for r in range(shards)]
            lists = [h.coarse_topn(q, nprobe, *coarse_slice(96, shards, r)) for r, h in enumerate(hs)]
            ids = DA.from_numpy(np.stack([l[0].astype(np.int64) for l in lists]))
            dist = DA.from_numpy(np.stack([l[1] for l in lists]))
            cnt = DA.from_numpy(np.stack([l[2].astype(np.int32) for l in lists]))
            pr, pd, pc = lancedb_amd.merge_topk(ids, dist, cnt, nprobe)
            probes = pr.numpy().astype(np.uint64)
            # the merged list IS the oracle's probe list (same coarse arithmetic, ties by partition id)
            co = o.coarse(q[0])
            assert sorted(probes[0].tolist()) == sorted(o.select_probes(co, nprobe).tolist())
            parts = [h.search_probes(q, probes, k=10) for h in hs]
            rid = DA.from_numpy(np.stack([p.rowids.astype(np.int64) for p in parts]))
            rd = DA.from_numpy(np.stack([p.distances for p in parts]))
            rc = DA.from_numpy(np.stack([p.counts.astype(np.int32) for p in parts]))
            mi, md, mc = lancedb_amd.merge_topk(rid, rd, rc, 10)
            assert (mi.numpy().astype(np.uint64) == exp[0]).all() and (md.numpy() == exp[1]).all()
            assert (mc.numpy().astype(np.uint32) == exp[2]).all()
        with pytest.raises(lancedb_amd.InvalidInput, match="not partitions"):
            hs[0].search_probes(q, np.full((41, 4), 1000, dtype=np.uint64), k=10)


def test_device_resident_io_and_device_index(oracle):
    DA = lancedb_amd.DeviceArray
    for m, dim in ((8, 64), (48, 192)):  # generic and skewed layouts
        s = train.synthetic_index(30000, dim, 16, m, seed=3)
        _, o = _both(oracle, s)
        g = lancedb_amd.IvfPqIndex(DA.from_numpy(s["centroids"]), DA.from_numpy(s["codebook"]), s["part_offsets"],
                                   DA.from_numpy(s["codes"]), DA.from_numpy(s["row_ids"].astype(np.int64)))
        q = np.random.default_rng(2).normal(size=(17, dim)).astype(np.float32)
        r = g.search(DA.from_numpy(q), k=10, nprobe_min=4, nprobe_max=4)
        g.sync()
        exp = o.search(q, k=10, nprobe_min=4, nprobe_max=4)
        assert (r.rowids.numpy().astype(np.uint64) == exp[0]).all()
        assert (r.distances.numpy() == exp[1]).all()
        assert (r.counts.numpy().astype(np.uint32) == exp[2]).all()


# ------------------------------------------------------------------ flat ----
@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
def test_flat_matches_oracle(oracle, metric):
    rng = np.random.default_rng(8)
    v = rng.normal(size=(70000, 48)).astype(np.float32)
    rid = rng.permutation(70000).astype(np.uint64)
    q = rng.normal(size=(5, 48)).astype(np.float32)
    f = lancedb_amd.FlatIndex(v, rid)
    mt = _abi.METRIC_NAMES[metric]
    for k in (1, 10, 100):
        _assert_same(f.search(q, k=k, metric=mt), oracle.flat_search(v, q, k=k, row_ids=rid, metric=mt))
    # half-open range and bf16 / f16 columns
    ids, dist, _, _ = oracle.flat_search(v, q, k=10, row_ids=rid, metric=mt)
    kw = dict(k=10, metric=mt, lower_bound=float(dist[0, 2]), upper_bound=float(dist[0, 6]))
    _assert_same(f.search(q, **kw), oracle.flat_search(v, q, row_ids=rid, **kw))
    bf = (v[:5000].view(np.uint32) >> 16).astype(np.uint16)
    fb = lancedb_amd.FlatIndex(bf, dtype=_abi.DTYPE_BF16)
    _assert_same(fb.search(q, k=10, metric=mt), oracle.flat_search(bf, q, k=10, dtype=_abi.DTYPE_BF16, metric=mt))
    h = v[:5000].astype(np.float16).view(np.uint16)
    fh = lancedb_amd.FlatIndex(h, dtype=_abi.DTYPE_F16)
    _assert_same(fh.search(q, k=10, metric=mt), oracle.flat_search(h, q, k=10, dtype=_abi.DTYPE_F16, metric=mt))


def test_flat_mfma_filter_is_exact_on_adversarial_columns(oracle):
    """The bf16 MFMA filter + exact re-rank must return exactly the exact sweep's
    rows: duplicates (every score ties -> candidate overflow -> exact re-scan),
    near-duplicates inside the bf16 rounding error, zero / NaN / huge rows, ragged
    batch sizes and dims that need padding."""
    rng = np.random.default_rng(17)
    n, dim = 20000, 80
    v = rng.normal(size=(n, dim)).astype(np.float32)
    v[100:4100] = v[100]                                  # 4000 exact duplicates
    v[5000:5200] = v[5000] + rng.normal(0, 1e-4, size=(200, dim)).astype(np.float32)  # inside bf16 eps
    v[6000] = 0.0                                         # zero norm (cosine -> NaN -> dropped)
    v[6001, 3] = np.nan
    v[6002] = 3e38                                        # squares overflow
    q = np.concatenate([v[[100, 5000, 6000, 6002]], rng.normal(size=(127, dim)).astype(np.float32)])
    f = lancedb_amd.FlatIndex(v)
    assert f.info()[1] == 1
    f.configure(path="filter")  # (by default the cheaper exact path per call: this table would be swept)
    for metric in ("l2", "cosine", "dot"):
        mt = _abi.METRIC_NAMES[metric]
        for k in (1, 10, 200):
            _assert_same(f.search(q, k=k, metric=mt), oracle.flat_search(v, q, k=k, metric=mt))
            assert f.info()[0] == 1
    # upper bound only stays on the filter path; a lower bound takes the exact sweep
    ids, dist, _, _ = oracle.flat_search(v, q, k=10)
    kw = dict(k=10, upper_bound=float(dist[5, 6]))
    _assert_same(f.search(q, **kw), oracle.flat_search(v, q, **kw))
    assert f.info()[0] == 1
    kw = dict(k=10, lower_bound=float(dist[5, 2]))
    _assert_same(f.search(q, **kw), oracle.flat_search(v, q, **kw))
    assert f.info()[0] == 2


def test_flat_mfma_bf16_column_c2_shape(oracle):
    """BASELINE.json configs[1] shape at reduced N: 768-d bf16 column, L2 and cosine,
    a 256-query batch; row ids permuted."""
    rng = np.random.default_rng(0x1A2CE)
    n, dim = 50000, 768
    v = rng.normal(size=(n, dim)).astype(np.float32)
    bf = (v.view(np.uint32) >> 16).astype(np.uint16)
    rid = rng.permutation(n).astype(np.uint64)
    q = rng.normal(size=(256, dim)).astype(np.float32)
    f = lancedb_amd.FlatIndex(bf, rid, dtype=_abi.DTYPE_BF16)
    f.configure(path="filter")
    for metric in ("l2", "cosine"):
        mt = _abi.METRIC_NAMES[metric]
        _assert_same(f.search(q, k=10, metric=mt),
                     oracle.flat_search(bf, q, k=10, row_ids=rid, dtype=_abi.DTYPE_BF16, metric=mt))
    assert f.info() == (1, 1)


def test_flat_reference_goldens_on_gpu():
    """The reference's own flat-search expectations, run through the HIP path."""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "flat_reference_cases.json")))
    for case in gold["cases"]:
        v = np.asarray(case["vectors"], dtype=np.float32)
        f = lancedb_amd.FlatIndex(v)
        r = f.search(np.asarray(case["query"], np.float32), k=case["k"], metric=_abi.METRIC_NAMES[case["metric"]])
        n = min(case["k"], len(v))
        assert r.counts[0] == n and r.rowids[0, :n].tolist() == case["expect_rowids"], case["name"]
        if "expect_dist" in case:
            np.testing.assert_allclose(r.distances[0, :n], case["expect_dist"], rtol=0, atol=case["atol"])


def test_c1_config_flat_100k_x_128(oracle):
    """BASELINE.json configs[0]: flat L2, 100k x 128 f32, one query."""
    rng = np.random.default_rng(0x1A2CE)
    v = rng.random(size=(100_000, 128), dtype=np.float32)
    q = rng.random(size=(1, 128), dtype=np.float32)
    f = lancedb_amd.FlatIndex(v)
    _assert_same(f.search(q, k=10), oracle.flat_search(v, q, k=10))


def test_vector_table_mirror_end_to_end(oracle):
    s = train.synthetic_index(9000, 16, 8, 4, seed=4)
    g, o = _both(oracle, s)
    raw = np.random.default_rng(0).normal(size=(9000, 16)).astype(np.float32)
    t = lancedb_amd.VectorTable(index=g, flat=lancedb_amd.FlatIndex(raw))
    q = np.random.default_rng(1).normal(size=(2, 16)).astype(np.float32)
    out = t.vector_search(q[0]).nprobes(4).limit(5).offset(2).execute()
    exp = o.search(q[:1], k=7, nprobe_min=4, nprobe_max=4)
    assert out["_rowid"].tolist() == exp[0][0, 2:7].tolist() and "query_index" not in out
    multi = t.vector_search(q).nprobes(4).limit(3).execute()  # table/query.rs:334-381
    assert multi["query_index"].tolist() == [0, 0, 0, 1, 1, 1]
    flat = t.vector_search(q[1]).bypass_vector_index().limit(4).execute()
    fi, fd, _, _ = oracle.flat_search(raw, q[1:], k=4)
    assert flat["_rowid"].tolist() == fi[0].tolist() and (flat["_distance"] == fd[0]).all()
    # analyze_plan (table/query.rs:105-112): the engine's own counters on the plan nodes
    text = t.vector_search(q[0]).nprobes(4).limit(5).analyze_plan()
    assert text.startswith("AnalyzeExec verbose=true, elapsed=") and "output_rows=5" in text
    scanned = sum(int(s["part_offsets"][p + 1] - s["part_offsets"][p]) for p in o.select_probes(o.coarse(q[0]), 4))
    assert f"rows_scanned={scanned}" in text and "partitions_ranked=4" in text and "elapsed_compute=" in text
    assert "KNNVectorDistance" in t.vector_search(q[1]).bypass_vector_index().limit(4).analyze_plan()


def test_c4_config_shape_nlist_65536_nprobe_128_sharded_coarse(oracle):
    """BASELINE.json configs[3] at a reduced row count: nlist = 65536, m = 96 x 8 bit,
    dim 768, nprobe = 128, partitions AND the coarse stage sharded 8 ways (two-phase
    search), merged with mi355_merge_topk == the unsharded oracle result."""
    from lancedb_amd.distributed import coarse_slice
    DA = lancedb_amd.DeviceArray
    nlist, dim, m, shards, nprobe = 65536, 768, 96, 8, 128
    s = train.synthetic_index(400_000, dim, nlist, m, seed=65, skew=1.0, empty_parts=1000)
    rng = np.random.default_rng(4)
    q = (s["centroids"][rng.integers(0, nlist, size=12)] + rng.normal(0, 0.3, size=(12, dim))).astype(np.float32)
    g, o = _both(oracle, s)
    exp = o.search(q, k=10, nprobe_min=nprobe, nprobe_max=nprobe)
    _assert_same(g.search(q, k=10, nprobe_min=nprobe, nprobe_max=nprobe), exp)
    hs = [lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"],
                                 shard_count=shards, shard_rank=r) for r in range(shards)]
    lists = [h.coarse_topn(q, nprobe, *coarse_slice(nlist, shards, r)) for r, h in enumerate(hs)]
    pr, _, _ = lancedb_amd.merge_topk(DA.from_numpy(np.stack([l[0].astype(np.int64) for l in lists])),
                                      DA.from_numpy(np.stack([l[1] for l in lists])),
                                      DA.from_numpy(np.stack([l[2].astype(np.int32) for l in lists])), nprobe)
    probes = pr.numpy().astype(np.uint64)
    parts = [h.search_probes(q, probes, k=10) for h in hs]
    mi, md, mc = lancedb_amd.merge_topk(DA.from_numpy(np.stack([p.rowids.astype(np.int64) for p in parts])),
                                        DA.from_numpy(np.stack([p.distances for p in parts])),
                                        DA.from_numpy(np.stack([p.counts.astype(np.int32) for p in parts])), 10)
    assert (mi.numpy().astype(np.uint64) == exp[0]).all() and (md.numpy() == exp[1]).all()
    assert (mc.numpy().astype(np.uint32) == exp[2]).all()


@pytest.mark.parametrize("raw_dtype", ["f32", "f16"])
def test_c5_config_shape_1536d_cosine_refine10(oracle, raw_dtype):
    """BASELINE.json configs[4] at a reduced row count: dim 1536, cosine, nprobe = 64,
    refine_factor = 10 with the raw-vector re-rank on the GPU (f32 and f16 raw columns)."""
    n, dim, nlist, m = 60_000, 1536, 256, 96
    s = train.synthetic_index(n, dim, nlist, m, seed=15, skew=0.7)
    rng = np.random.default_rng(8)
    raw = rng.normal(size=(n, dim)).astype(np.float32)
    dt = _abi.DTYPE_F32
    if raw_dtype == "f16":
        raw, dt = raw.astype(np.float16).view(np.uint16), _abi.DTYPE_F16
    g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"],
                               raw_vectors=raw, metric="cosine", raw_dtype=dt)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"],
                           raw_vectors=raw, metric="cosine", raw_dtype=dt)
    q = rng.normal(size=(24, dim)).astype(np.float32)
    for kw in (dict(k=10, nprobe_min=64, nprobe_max=64, refine_factor=10), dict(k=10, nprobe_min=64, nprobe_max=64)):
        _assert_same(g.search(q, **kw), o.search(q, **kw))


GEMM_VARIANTS = {"128": _abi.FLAT_GEMM_128, "256": _abi.FLAT_GEMM_256,
                 "8phase": _abi.FLAT_GEMM_8PHASE, "8phase_ref": _abi.FLAT_GEMM_8PHASE_REF}


@pytest.mark.parametrize("tile", sorted(GEMM_VARIANTS))
def test_flat_mfma_persistent_workgroups_walk_many_tiles(oracle, tile):
    """A grid of 8 workgroups (one per XCD) walks every tile of the column: the cross-tile
    path of the flat GEMM (next tile's first stages issued under the last k-steps, ragged last
    row tile, virtual blocks that map to no row tile) against the exact sweep."""
    rng = np.random.default_rng(99)
    n, dim = 9000 + 37, 136  # 36 row tiles of 256 (last one ragged), dim padded to 192 -> 3 k-tiles
    v = rng.normal(size=(n, dim)).astype(np.float32)
    q = rng.normal(size=(300, dim)).astype(np.float32)  # 2 query tiles of 256 / 3 of 128
    f = lancedb_amd.FlatIndex(v)
    f.configure(gemm_variant=GEMM_VARIANTS[tile], grid_workgroups=8)
    for metric in ("l2", "cosine", "dot"):
        mt = _abi.METRIC_NAMES[metric]
        _assert_same(f.search(q, k=10, metric=mt), oracle.flat_search(v, q, k=10, metric=mt))
        assert f.info()[0] == 1
        assert f.stats()["fallback_queries"] == 0  # random data: the filter itself must have done the work
    # a single k-tile per row tile (dim <= 64): the first stage of the next tile is the only stage
    # (the 8-phase walk needs two k-tiles and hands such columns to the two-barrier kernel)
    v1, q1 = np.ascontiguousarray(v[:, :40]), np.ascontiguousarray(q[:, :40])
    f1 = lancedb_amd.FlatIndex(v1)
    f1.configure(gemm_variant=GEMM_VARIANTS[tile], grid_workgroups=8)
    _assert_same(f1.search(q1, k=5), oracle.flat_search(v1, q1, k=5))
    assert f1.info()[0] == 1


@pytest.mark.parametrize("variant", ["8phase", "8phase_ref"])
@pytest.mark.parametrize("grid", [0, 1, 16])
def test_flat_mfma_eight_phase_schedule(oracle, variant, grid):
    """The persistent 8-phase schedule against the exact sweep: ragged last row tile, 2 / 3 / 12
    k-tiles, three metrics, one workgroup per tile (grid 1), per CU slot (0) and a 16-workgroup
    walk; every search repeated (a staging race would come and go)."""
    rng = np.random.default_rng(123)
    for n, dim in ((9000 + 37, 100), (5000, 136), (6000, 768)):
        v = rng.normal(size=(n, dim)).astype(np.float32)
        q = rng.normal(size=(300, dim)).astype(np.float32)
        f = lancedb_amd.FlatIndex(v)
        f.configure(gemm_variant=GEMM_VARIANTS[variant], grid_workgroups=grid)
        for metric in ("l2", "cosine", "dot"):
            mt = _abi.METRIC_NAMES[metric]
            exp = oracle.flat_search(v, q, k=10, metric=mt)
            for _ in range(3):
                _assert_same(f.search(q, k=10, metric=mt), exp)
                assert f.stats()["fallback_queries"] == 0  # a garbage filter would hide behind the exact re-scan
            assert f.info()[0] == 1


def test_flat_mfma_eight_phase_reference_epilogue_matches_two_barrier_kernel_bit_for_bit(oracle):
    """Same operands, same MFMA k-order, same epilogue arithmetic: the 8-phase schedule with the
    reference epilogue must produce the group-minimum matrix of the two-barrier kernel exactly
    (order-independent checksum), whatever the grid."""
    rng = np.random.default_rng(7)
    v = rng.normal(size=(20000 + 11, 320)).astype(np.float32)
    q = rng.normal(size=(512, 320)).astype(np.float32)
    f = lancedb_amd.FlatIndex(v)
    sums = {}
    for name, var, grid in (("256", _abi.FLAT_GEMM_256, 0), ("8ref", _abi.FLAT_GEMM_8PHASE_REF, 0),
                            ("8ref_g1", _abi.FLAT_GEMM_8PHASE_REF, 1), ("8ref_g24", _abi.FLAT_GEMM_8PHASE_REF, 24),
                            ("8ref_g256", _abi.FLAT_GEMM_8PHASE_REF, 256)):
        for metric in ("l2", "cosine", "dot"):
            f.configure(gemm_variant=var, grid_workgroups=grid, checksum=True)
            f.search(q, k=10, metric=_abi.METRIC_NAMES[metric])
            sums[(name, metric)] = f.checksum()
    for metric in ("l2", "cosine", "dot"):
        assert len({sums[(n, metric)] for n in ("256", "8ref", "8ref_g1", "8ref_g24", "8ref_g256")}) == 1, sums


@pytest.mark.parametrize("variant", sorted(GEMM_VARIANTS))
def test_flat_gemm_variants_on_a_chip_filling_grid(variant):
    """Every schedule on a column large enough for every CU to walk many tiles (and, with grid 1,
    for thousands of one-tile workgroups): identical results across variants and no candidate-list
    overflow (the exact re-scan would mask a wrong filter; checked against the exact sweep on a
    sample of queries)."""
    rng = np.random.default_rng(5)
    n, dim = 400_000, 256
    v = rng.normal(size=(n, dim)).astype(np.float32)
    q = rng.normal(size=(512, dim)).astype(np.float32)
    f = lancedb_amd.FlatIndex(v)
    f.configure(gemm_variant=_abi.FLAT_GEMM_256)
    ref = f.search(q, k=10)
    assert f.stats()["fallback_queries"] == 0
    for grid in (0, 1, 256):
        f.configure(gemm_variant=GEMM_VARIANTS[variant], grid_workgroups=grid)
        for _ in range(2):
            got = f.search(q, k=10)
            assert f.stats()["fallback_queries"] == 0, (variant, grid)
            assert (got.rowids == ref.rowids).all() and (got.distances == ref.distances).all()


def test_flat_adversarial_columns_on_the_eight_phase_schedule(oracle):
    """The fast epilogue's "never filter" rule (non-finite sum of a group's terms) on duplicates,
    near-duplicates inside the bf16 error, zero / NaN / 3e38 rows and a zero query."""
    rng = np.random.default_rng(17)
    n, dim = 20000, 128
    v = rng.normal(size=(n, dim)).astype(np.float32)
    v[100:4100] = v[100]
    v[5000:5200] = v[5000] + rng.normal(0, 1e-4, size=(200, dim)).astype(np.float32)
    v[6000] = 0.0
    v[6001, 3] = np.nan
    v[6002] = 3e38
    q = np.concatenate([v[[100, 5000, 6000, 6002]], np.zeros((1, dim), np.float32),
                        rng.normal(size=(251, dim)).astype(np.float32)])
    f = lancedb_amd.FlatIndex(v)
    for variant, grid in ((_abi.FLAT_GEMM_8PHASE, 0), (_abi.FLAT_GEMM_8PHASE, 8)):
        f.configure(gemm_variant=variant, grid_workgroups=grid)
        for metric in ("l2", "cosine", "dot"):
            mt = _abi.METRIC_NAMES[metric]
            for k in (1, 10, 200):
                _assert_same(f.search(q, k=k, metric=mt), oracle.flat_search(v, q, k=k, metric=mt))
                assert f.info()[0] == 1


def test_flat_takes_the_cheaper_exact_path_per_call_and_both_return_the_same(oracle):
    """Flat search (python/python/lancedb/query.py:1365-1370) has two exact paths; by default the call's size picks one
    (csrc/ann_flat.hip, run_flat_search_device): a single query sweeps, a batch of hundreds runs the MFMA filter.  The
    `path` pins (`MI355_FLAT_FORCE_FILTER` / `_SWEEP`) and the default must agree bit for bit, host and device I/O."""
    rng = np.random.default_rng(23)
    v = rng.normal(size=(20000, 64)).astype(np.float32)
    q = rng.normal(size=(300, 64)).astype(np.float32)
    f = lancedb_amd.FlatIndex(v)
    for nq, default_path in ((1, 2), (3, 2), (300, 1)):
        exp = oracle.flat_search(v, q[:nq], k=10)
        for path, took in ((None, default_path), ("filter", 1), ("sweep", 2)):
            f.configure(path=path)
            _assert_same(f.search(q[:nq], k=10), exp)
            assert f.info()[0] == took, (nq, path)
    import ctypes as C

    from lancedb_amd._lib import lib
    both = _abi.FLAT_FORCE_FILTER | _abi.FLAT_FORCE_SWEEP  # the two pins exclude each other
    assert lib().mi355_flat_configure(f._h, C.c_uint32(0), C.c_uint32(0), C.c_uint32(both)) == _abi.ERR_INVALID_INPUT


@pytest.mark.gpu
@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_blocked_coarse_kernel_partial_tiles(oracle, metric):
    """Batches whose coarse stage launches at least two 128 x 128 tiles per CU run the register-blocked f32 MFMA kernel
    (k_coarse_mfma2, csrc/kernels_ivfpq.h): nlist and the batch are not multiples of 128, dim is a multiple of 4 but not
    of the 32-deep k stage.  A wrong coarse score changes a probe list, and with it ids or distances: everything `==`."""
    nlist, dim, m, nq, nprobe = 33001, 100, 25, 300, 6
    s = train.synthetic_index(150_000, dim, nlist, m, seed=91, skew=0.8, empty_parts=500)
    rng = np.random.default_rng(17)
    q = (s["centroids"][rng.integers(0, nlist, size=nq)] + rng.normal(0, 0.4, size=(nq, dim))).astype(np.float32)
    g, o = _both(oracle, s, metric=metric)
    _assert_same(g.search(q, k=10, nprobe_min=nprobe, nprobe_max=nprobe), o.search(q, k=10, nprobe_min=nprobe, nprobe_max=nprobe))


def test_blocked_coarse_kernel_in_the_coarse_slice_path(oracle):
    """mi355_coarse_topn (the sharded coarse stage of SURVEY.md §8e) over a centroid slice large enough for the register-blocked
    kernel (two 128 x 128 tiles per CU), slice bounds and batch not multiples of 128: scores and ids `==` the oracle's."""
    nlist, dim, m, nq, n_sel, lo, hi = 40000, 64, 16, 600, 9, 1237, 1237 + 17003
    s = train.synthetic_index(60_000, dim, nlist, m, seed=92, skew=0.5, empty_parts=100)
    rng = np.random.default_rng(18)
    q = (s["centroids"][rng.integers(lo, hi, size=nq)] + rng.normal(0, 0.4, size=(nq, dim))).astype(np.float32)
    g, o = _both(oracle, s)
    ids, dist, cnt = g.coarse_topn(q, n_sel, lo, hi)
    assert (np.asarray(cnt) == n_sel).all()
    for i in range(0, nq, 7):
        co = np.asarray(o.coarse(q[i]))[lo:hi]
        order = np.lexsort((np.arange(lo, hi), co))[:n_sel]
        got = sorted(zip(np.asarray(dist[i]).tolist(), np.asarray(ids[i]).astype(np.int64).tolist()))
        assert got == sorted(zip(co[order].tolist(), (order + lo).tolist())), i
