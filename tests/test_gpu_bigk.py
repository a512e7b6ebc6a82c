"""No cap on `limit` or `refine_factor` (the reference bounds neither:
rust/lancedb/src/query.rs:818-907, :1302-1332; its published chart uses refine_factor 30-50):
k * refine_factor beyond the 256 rows one selection pass holds runs the SAME selectors in
passes (device_common.h WaveTopK floor, kernels' MULTI instantiations).  Bit-exact against the
oracle on both scan kernels, the merge, the refine stage, the flat paths and merge_topk."""
import numpy as np
import pytest

import lancedb_amd
from lancedb_amd import _abi
from oracle import train

pytestmark = pytest.mark.gpu


def _same(got, exp):
    ids, dist, cnt, st = exp
    assert st == 0
    assert (got.counts == cnt).all()
    assert (got.rowids == ids).all()
    assert (got.distances == dist).all()


@pytest.mark.parametrize("m,dim,scan", [(8, 32, _abi.SCAN_PAIR), (32, 128, _abi.SCAN_SKEW)])
def test_ivfpq_limits_beyond_one_selection_pass(oracle, m, dim, scan):
    s = train.synthetic_index(60000, dim, 16, m, seed=m, skew=0.7, empty_parts=1)
    g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"],
                               generic_scan=scan == _abi.SCAN_PAIR)  # (m = 8 would otherwise be padded onto the production scan)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    q = np.random.default_rng(2).normal(size=(7, dim)).astype(np.float32)
    for k, nprobe in ((257, 4), (300, 4), (1000, 6), (1000, 1), (5000, 16)):
        _same(g.search(q, k=k, nprobe_min=nprobe, nprobe_max=nprobe), o.search(q, k=k, nprobe_min=nprobe, nprobe_max=nprobe))
        assert g.stats()["scan_variant"] == scan
    # a range that leaves fewer rows than asked for, and the maximum_nprobes second pass at k = 600
    ids, dist, cnt, _ = o.search(q, k=1000, nprobe_min=4, nprobe_max=4)
    kw = dict(k=600, nprobe_min=2, nprobe_max=9, upper_bound=float(dist[0, 400]))
    _same(g.search(q, **kw), o.search(q, **kw))


def test_ivfpq_ties_across_selection_passes(oracle):
    """Every row of a partition has the same code: 4000 equal distances, so the pass floor must
    separate rows by row id alone — also exactly at the pass boundaries (ranks 255 | 256)."""
    for m, dim in ((8, 32), (32, 128)):
        s = train.synthetic_index(12000, dim, 4, m, seed=3)
        s["codes"][:] = s["codes"][0]
        s["row_ids"] = np.random.default_rng(0).permutation(12000).astype(np.uint64) + (1 << 40)
        g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
        o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
        q = np.random.default_rng(4).normal(size=(3, dim)).astype(np.float32)
        for k in (256, 257, 512, 700):
            _same(g.search(q, k=k, nprobe_min=2, nprobe_max=2), o.search(q, k=k, nprobe_min=2, nprobe_max=2))


@pytest.mark.parametrize("metric", ["l2", "cosine"])
def test_refine_factors_of_the_reference_chart(oracle, metric):
    """refine_factor 30 and 50 at k = 10 and k = 100: kk up to 5000."""
    rng = np.random.default_rng(21)
    cent = rng.normal(size=(64, 64)).astype(np.float32) * 3
    x = (cent[rng.integers(0, 64, size=40000)] + rng.normal(size=(40000, 64))).astype(np.float32)
    t = train.train_ivfpq(x, nlist=32, m=16, metric=metric, iters=4)
    q = (cent[rng.integers(0, 64, size=12)] + rng.normal(size=(12, 64))).astype(np.float32)
    for generic in (True, False):  # m = 16 has no skewed kernel; 32 below does
        g = lancedb_amd.IvfPqIndex(t["centroids"], t["codebook"], t["part_offsets"], t["codes"], t["row_ids"],
                                   raw_vectors=t["raw"], metric=metric, generic_scan=generic)
        o = oracle.OracleIndex(t["centroids"], t["codebook"], t["part_offsets"], t["codes"], t["row_ids"],
                               raw_vectors=t["raw"], metric=metric)
        for k, rf in ((10, 30), (10, 50), (100, 50), (300, 2)):
            kw = dict(k=k, nprobe_min=8, nprobe_max=8, refine_factor=rf)
            _same(g.search(q, **kw), o.search(q, **kw))
    t2 = train.train_ivfpq(x, nlist=16, m=32, metric=metric, iters=3)
    g = lancedb_amd.IvfPqIndex(t2["centroids"], t2["codebook"], t2["part_offsets"], t2["codes"], t2["row_ids"],
                               raw_vectors=t2["raw"], metric=metric)
    o = oracle.OracleIndex(t2["centroids"], t2["codebook"], t2["part_offsets"], t2["codes"], t2["row_ids"],
                           raw_vectors=t2["raw"], metric=metric)
    assert g.search(q, k=10, nprobe_min=4, nprobe_max=4, refine_factor=50).counts.min() == 10
    assert g.stats()["scan_variant"] == _abi.SCAN_SKEW
    for k, rf in ((10, 50), (40, 30)):
        kw = dict(k=k, nprobe_min=4, nprobe_max=4, refine_factor=rf)
        _same(g.search(q, **kw), o.search(q, **kw))


def test_flat_limits_beyond_one_selection_pass(oracle):
    rng = np.random.default_rng(8)
    v = rng.normal(size=(30000, 72)).astype(np.float32)
    v[500:1500] = v[500]  # a thousand exact ties
    q = np.concatenate([v[[500]], rng.normal(size=(140, 72)).astype(np.float32)])
    f = lancedb_amd.FlatIndex(v)
    f.configure(path="filter")  # (a table this small is swept by default: the cheaper of the two exact paths)
    for metric in ("l2", "cosine", "dot"):
        mt = _abi.METRIC_NAMES[metric]
        for k in (300, 1000, 2500):
            _same(f.search(q, k=k, metric=mt), oracle.flat_search(v, q, k=k, metric=mt))
            assert f.info()[0] == 1  # MFMA filter + re-rank in passes
        # lower-bounded range -> the exact sweep, in passes
        ids, dist, cnt, _ = oracle.flat_search(v, q[:9], k=400, metric=mt)
        kw = dict(k=700, metric=mt, lower_bound=float(dist[1, 50]), upper_bound=float(dist[1, 390]))
        _same(f.search(q[:9], **kw), oracle.flat_search(v, q[:9], **kw))
        assert f.info()[0] == 2
    small = lancedb_amd.FlatIndex(v[:900])  # no filter data: exact sweep; k beyond the column
    _same(small.search(q[:5], k=1200), oracle.flat_search(v[:900], q[:5], k=1200))


def test_merge_topk_beyond_one_selection_pass(oracle):
    DA = lancedb_amd.DeviceArray
    rng = np.random.default_rng(5)
    n_lists, nq, k = 5, 9, 700
    ids = rng.permutation(n_lists * nq * k).astype(np.uint64).reshape(n_lists, nq, k)
    dist = np.sort(rng.integers(0, 300, size=(n_lists, nq, k)).astype(np.float32), axis=2)  # many cross-list ties
    cnt = rng.integers(0, k + 1, size=(n_lists, nq)).astype(np.uint32)
    g_ids, g_dist, g_cnt = lancedb_amd.merge_topk(DA.from_numpy(ids.view(np.int64)), DA.from_numpy(dist), DA.from_numpy(cnt.view(np.int32)), k)
    e_ids, e_dist, e_cnt = oracle.merge_topk(ids, dist, cnt, k)
    assert (g_cnt.numpy().view(np.uint32) == e_cnt).all()
    assert (g_ids.numpy().view(np.uint64) == e_ids).all() and (g_dist.numpy() == e_dist).all()


@pytest.mark.parametrize("m,dim", [(32, 128), (96, 192)])
def test_long_lists_on_sixteen_waves(oracle, m, dim):
    """k * refine_factor > 128 on the production scan: sixteen waves whose candidate lists (192 rows)
    are SHORTER than the 256 rows a pass selects, kept short by the workgroup-shared threshold
    (k_scan_skew OPT / QSHARE).  Random row order (the optimistic pass succeeds) and the adversarial one: the best rows
    of a partition all sit in the tiles ONE wave scans, so its list overflows and the work item is
    redone in passes of 128 rows — same results either way, bit-exact against the oracle."""
    rng = np.random.default_rng(m)
    s = train.synthetic_index(50000, dim, 6, m, seed=m + 1, skew=0.3)
    po = s["part_offsets"].astype(np.int64)
    q = rng.normal(size=(5, dim)).astype(np.float32)
    raw = rng.normal(size=(50000, dim)).astype(np.float32)

    def check(codes):
        g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], codes, s["row_ids"], raw_vectors=raw)
        o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], codes, s["row_ids"], raw_vectors=raw)
        for kw in (dict(k=129, nprobe_min=3, nprobe_max=3), dict(k=200, nprobe_min=6, nprobe_max=6),
                   dict(k=250, nprobe_min=2, nprobe_max=2), dict(k=256, nprobe_min=6, nprobe_max=6),
                   dict(k=10, nprobe_min=4, nprobe_max=4, refine_factor=25), dict(k=192, nprobe_min=1, nprobe_max=1),
                   dict(k=600, nprobe_min=1, nprobe_max=1), dict(k=10, nprobe_min=3, nprobe_max=3, refine_factor=50),
                   dict(k=240, nprobe_min=2, nprobe_max=6, upper_bound=float(o.search(q, k=300, nprobe_min=6, nprobe_max=6)[1][0, 150]))):
            _same(g.search(q, **kw), o.search(q, **kw))
        assert g.stats()["scan_variant"] == _abi.SCAN_SKEW

    check(s["codes"])
    # adversarial: in every partition, the rows of the tiles that unit 0 scans (tile % 32 in {0, 1}: two
    # chains per wave) get the codes of the partition's first row with a few bytes varied — hundreds of
    # near-equal best rows for a query near that row's reconstruction, all in one wave's list
    codes = s["codes"].copy()
    for p in range(len(po) - 1):
        n_p = po[p + 1] - po[p]
        rows = np.arange(n_p)
        crowd = rows[((rows // 64) % 32) < 2]
        base = codes[po[p]].copy()
        codes[po[p] + crowd] = base
        codes[po[p] + crowd, rng.integers(0, m, size=len(crowd))] = rng.integers(0, 256, size=len(crowd)).astype(np.uint8)
    # queries = reconstructions of each partition's first row (so the crowd is the near set)
    cb = s["codebook"]
    recon = []
    for p in range(min(5, len(po) - 1)):
        c0 = codes[po[p]]
        recon.append(s["centroids"][p] + np.concatenate([cb[j, c0[j]] for j in range(m)]))
    q[:len(recon)] = np.asarray(recon, dtype=np.float32)
    check(codes)
