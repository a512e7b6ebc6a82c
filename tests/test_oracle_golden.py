"""Pins the CPU oracle: reference golden expectations for flat search, and
self-consistency of the (unpinned) IVF-PQ restatement against the pinned flat
path.  CPU only."""
import json
import os

import numpy as np
import pytest

from lancedb_amd import _abi
from oracle import train

GOLD = os.path.join(os.path.dirname(__file__), "golden", "flat_reference_cases.json")


def _cases():
    with open(GOLD) as f:
        return json.load(f)


@pytest.mark.parametrize("case", _cases()["cases"], ids=lambda c: c["name"])
def test_flat_reference_goldens(oracle, case):
    v = np.asarray(case["vectors"], dtype=np.float32)
    q = np.asarray(case["query"], dtype=np.float32)
    ids, dist, cnt, st = oracle.flat_search(v, q, k=case["k"], metric=_abi.METRIC_NAMES[case["metric"]])
    assert st == 0
    n = min(case["k"], len(v))
    assert cnt[0] == n
    assert ids[0, :n].tolist() == case["expect_rowids"]
    if "expect_dist" in case:
        np.testing.assert_allclose(dist[0, :n], case["expect_dist"], rtol=0, atol=case["atol"])
    if case["metric"] == "cosine":
        assert 0 <= dist[0, 0] <= 1  # test_query.py:1014


def test_flat_distance_range_boundaries(oracle):
    rc = _cases()["range_cases"]  # test_query.py:655-675
    v = np.asarray(rc["vectors"], dtype=np.float32)
    q = np.asarray(rc["query"], dtype=np.float32)
    _, d, cnt, _ = oracle.flat_search(v, q, k=10)
    lo, hi = float(d[0, 0]), float(d[0, 1])
    assert cnt[0] == 2 and (lo, hi) == (5.0, 25.0)  # squared L2 of [1,2],[3,4] from the origin
    val = {"min": lo, "max": hi}
    for chk in rc["checks"]:
        kw = {}
        if "upper" in chk:
            kw["upper_bound"] = val[chk["upper"]]
        if "lower" in chk:
            kw["lower_bound"] = val[chk["lower"]]
        _, dd, c, _ = oracle.flat_search(v, q, k=10, **kw)
        assert c[0] == chk["expect_count"]
        exp = {"min": [lo], "max": [hi], "both": [lo, hi]}.get(chk.get("expect"), [])
        assert dd[0, :c[0]].tolist() == exp


def test_flat_matches_numpy_float64(oracle):
    rng = np.random.default_rng(1)
    v = rng.normal(size=(500, 24)).astype(np.float32)
    q = rng.normal(size=(3, 24)).astype(np.float32)
    for metric, ref in (("l2", lambda a, b: ((a - b) ** 2).sum(-1)),
                        ("dot", lambda a, b: 1 - (a * b).sum(-1)),
                        ("cosine", lambda a, b: 1 - (a * b).sum(-1) / np.linalg.norm(a, axis=-1) / np.linalg.norm(b, axis=-1))):
        ids, dist, cnt, st = oracle.flat_search(v, q, k=7, metric=_abi.METRIC_NAMES[metric])
        assert st == 0 and (cnt == 7).all()
        for i in range(3):
            full = ref(v.astype(np.float64), q[i].astype(np.float64)[None, :])
            order = np.lexsort((np.arange(500), full))[:7]
            assert ids[i].tolist() == order.tolist()
            np.testing.assert_allclose(dist[i], full[order], rtol=1e-5, atol=1e-6)


def test_flat_ties_break_by_rowid_and_nan_dropped(oracle):
    v = np.zeros((6, 4), dtype=np.float32)
    v[:, 0] = [1, 1, 1, 1, 1, 1]
    rid = np.array([50, 10, 40, 20, 30, 60], dtype=np.uint64)
    ids, dist, cnt, _ = oracle.flat_search(v, np.zeros(4, np.float32), k=4, row_ids=rid)
    assert ids[0].tolist() == [10, 20, 30, 40] and (dist[0] == 1.0).all()
    # cosine against an all-zero vector is undefined -> NULL -> dropped (lib.rs:247-249)
    v2 = np.array([[0, 0], [1, 0], [0, 2]], dtype=np.float32)
    ids, dist, cnt, _ = oracle.flat_search(v2, np.array([1, 0], np.float32), k=3, metric=_abi.METRIC_COSINE)
    assert cnt[0] == 2 and ids[0, :2].tolist() == [1, 2]
    assert ids[0, 2] == _abi.UINT64_MAX and np.isinf(dist[0, 2])


def test_flat_bf16_and_f16_widen_exactly(oracle):
    rng = np.random.default_rng(5)
    v32 = rng.normal(size=(64, 16)).astype(np.float32)
    bf = (v32.view(np.uint32) >> 16).astype(np.uint16)
    back = (bf.astype(np.uint32) << 16).view(np.float32)
    q = rng.normal(size=(2, 16)).astype(np.float32)
    a = oracle.flat_search(bf, q, k=5, dtype=_abi.DTYPE_BF16)
    b = oracle.flat_search(back, q, k=5)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
    h = v32.astype(np.float16)
    a = oracle.flat_search(h.view(np.uint16), q, k=5, dtype=_abi.DTYPE_F16)
    b = oracle.flat_search(h.astype(np.float32), q, k=5)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all()


def _lossless_index(rng, n, dim, nlist, m, metric):
    """Vectors that the PQ represents exactly: centroid + codebook entries."""
    dsub = dim // m
    cen = rng.integers(-4, 5, size=(nlist, dim)).astype(np.float32)
    if metric == "dot":
        cen[:] = 0
    cb = rng.integers(-3, 4, size=(m, 256, dsub)).astype(np.float32) * 0.5
    assign = np.sort(rng.integers(0, nlist, size=n))
    codes = rng.integers(0, 256, size=(n, m), dtype=np.uint8)
    vec = cen[assign].copy()
    for j in range(m):
        vec[:, j * dsub:(j + 1) * dsub] += cb[j][codes[:, j]]
    po = np.zeros(nlist + 1, dtype=np.uint64)
    po[1:] = np.cumsum(np.bincount(assign, minlength=nlist))
    return cen, cb, po, codes, vec


@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_ivfpq_lossless_equals_flat(oracle, metric):
    """IVF-PQ with nprobe = nlist over exactly-representable vectors must rank
    like the (reference-pinned) flat path; small-integer data keeps f32 exact."""
    rng = np.random.default_rng(7)
    cen, cb, po, codes, vec = _lossless_index(rng, 800, 16, 8, 4, metric)
    rid = rng.permutation(800).astype(np.uint64)
    ix = oracle.OracleIndex(cen, cb, po, codes, row_ids=rid, metric=metric)
    q = rng.integers(-5, 6, size=(5, 16)).astype(np.float32)
    ids, dist, cnt, st = ix.search(q, k=10, nprobe_min=8, nprobe_max=8)
    assert st == 0
    fids, fdist, fcnt, _ = oracle.flat_search(vec, q, k=10, row_ids=rid, metric=_abi.METRIC_NAMES[metric])
    assert (ids == fids).all()
    np.testing.assert_allclose(dist, fdist, rtol=1e-6, atol=1e-5)


def test_ivfpq_transposed_layout_equals_row_major(oracle):
    s = train.synthetic_index(3000, 32, 16, 8, seed=3, empty_parts=2)
    q = np.random.default_rng(0).normal(size=(4, 32)).astype(np.float32)
    a = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    t = train.to_part_transposed(s["codes"], s["part_offsets"])
    b = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], t, s["row_ids"],
                           codes_layout=_abi.CODES_PART_TRANSPOSED)
    ra, rb = a.search(q, k=10, nprobe_min=5, nprobe_max=5), b.search(q, k=10, nprobe_min=5, nprobe_max=5)
    assert (ra[0] == rb[0]).all() and (ra[1] == rb[1]).all() and (ra[2] == rb[2]).all()


def test_ivfpq_stagewise_matches_python_restatement(oracle):
    """The C stages against a line-by-line numpy/float32 restatement of the contract."""
    s = train.synthetic_index(600, 16, 6, 4, seed=11)
    ix = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    q = np.random.default_rng(2).normal(size=16).astype(np.float32)
    f = np.float32

    def fma(a, b, c):  # exact fused multiply-add in float32 via float64 (24+24 < 53 bits)
        return f(np.float64(a) * np.float64(b) + np.float64(c))

    def cdot(a, b):
        acc = f(0)
        for x, y in zip(a, b):
            acc = fma(x, y, acc)
        return acc

    coarse = ix.coarse(q)
    for p in range(6):
        c = s["centroids"][p]
        exp = fma(f(-2), cdot(q, c), f(cdot(q, q) + cdot(c, c)))
        assert coarse[p] == exp
    probes = ix.select_probes(coarse, 3)
    assert probes.tolist() == np.lexsort((np.arange(6), coarse))[:3].tolist()
    p = int(probes[0])
    lut = ix.build_lut(q, p)
    r = (q - s["centroids"][p]).astype(f)
    for j in (0, 3):
        for c in (0, 17, 255):
            acc = f(0)
            for t in range(4):
                d = f(r[j * 4 + t] - s["codebook"][j, c, t])
                acc = fma(d, d, acc)
            assert lut[j, c] == acc
    adc = ix.adc_partition(lut, p)
    o, e = int(s["part_offsets"][p]), int(s["part_offsets"][p + 1])
    for i in range(0, e - o, max(1, (e - o) // 7)):
        acc = f(0)
        for j in range(4):
            acc = f(acc + lut[j, s["codes"][o + i, j]])
        assert adc[i] == acc


def test_ivfpq_semantics(oracle):
    s = train.synthetic_index(4000, 32, 16, 8, seed=5, empty_parts=3)
    ix = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    q = np.random.default_rng(1).normal(size=(6, 32)).astype(np.float32)
    # more probes can only improve (or keep) the k-th distance
    _, d4, _, _ = ix.search(q, k=10, nprobe_min=4, nprobe_max=4)
    ids_all, d_all, c_all, _ = ix.search(q, k=10, nprobe_min=16, nprobe_max=16)
    assert (d_all <= d4 + 0).all() and (c_all == 10).all()
    # sorted by (distance, rowid)
    assert (np.diff(d_all, axis=1) >= 0).all()
    # distance_range is half-open [lower, upper) on the returned distances
    lo, hi = float(d_all[0, 2]), float(d_all[0, 7])
    ids_r, d_r, c_r, _ = ix.search(q[:1], k=10, nprobe_min=16, nprobe_max=16, lower_bound=lo, upper_bound=hi)
    assert c_r[0] == 5 and d_r[0, 0] == lo and (d_r[0, :5] < hi).all()
    assert ids_r[0, :5].tolist() == ids_all[0, 2:7].tolist()
    # k larger than the index: everything, then padding
    small = train.synthetic_index(7, 8, 2, 2, seed=1)
    six = oracle.OracleIndex(small["centroids"], small["codebook"], small["part_offsets"], small["codes"])
    i2, d2, c2, _ = six.search(np.zeros(8, np.float32), k=10, nprobe_min=2, nprobe_max=2)
    assert c2[0] == 7 and (i2[0, 7:] == _abi.UINT64_MAX).all() and np.isinf(d2[0, 7:]).all()
    # maximum_nprobes: excess partitions searched only when the first pass is short
    i3, d3, c3, _ = six.search(np.zeros(8, np.float32), k=10, nprobe_min=1, nprobe_max=2)
    assert c3[0] == 7
    # parameter validation mirrors query.rs:1232-1275
    assert ix.search(q, k=10, nprobe_min=0, nprobe_max=4)[3] == _abi.ERR_INVALID_INPUT
    assert ix.search(q, k=10, nprobe_min=5, nprobe_max=4)[3] == _abi.ERR_INVALID_INPUT
    assert ix.search(q, k=10, metric=_abi.METRIC_COSINE)[3] == _abi.ERR_INVALID_INPUT


def test_ivfpq_cosine_is_half_l2_of_unit_vectors(oracle):
    rng = np.random.default_rng(9)
    x = rng.normal(size=(1500, 16)).astype(np.float32)
    t = train.train_ivfpq(x, nlist=8, m=4, metric="cosine", iters=4)
    ixc = oracle.OracleIndex(t["centroids"], t["codebook"], t["part_offsets"], t["codes"], t["row_ids"],
                             raw_vectors=t["raw"], metric="cosine")
    ixl = oracle.OracleIndex(t["centroids"], t["codebook"], t["part_offsets"], t["codes"], t["row_ids"], metric="l2")
    q = rng.normal(size=(3, 16)).astype(np.float32)
    ic, dc, _, _ = ixc.search(q, k=5, nprobe_min=8, nprobe_max=8)
    qn = np.stack([ixc.preprocess(v) for v in q])
    il, dl, _, _ = ixl.search(qn, k=5, nprobe_min=8, nprobe_max=8)
    assert (ic == il).all() and (dc == dl * np.float32(0.5)).all()
    # refine replaces approximate distances by exact ones and re-sorts (query.rs:1313-1317)
    ir, dr, cr, _ = ixc.search(q, k=5, nprobe_min=8, nprobe_max=8, refine_factor=4)
    pos = {int(r): i for i, r in enumerate(t["row_ids"])}
    for qi in range(3):
        for j in range(int(cr[qi])):
            v = t["raw"][pos[int(ir[qi, j])]]
            exp = 1 - np.dot(q[qi].astype(np.float64), v) / np.linalg.norm(q[qi].astype(np.float64)) / np.linalg.norm(v.astype(np.float64))
            assert abs(dr[qi, j] - exp) < 1e-5
        assert (np.diff(dr[qi, :cr[qi]]) >= 0).all()


def test_recall_of_trained_index(oracle):
    rng = np.random.default_rng(21)
    cent = rng.normal(size=(32, 32)).astype(np.float32) * 3
    x = (cent[rng.integers(0, 32, size=6000)] + rng.normal(size=(6000, 32))).astype(np.float32)
    t = train.train_ivfpq(x, nlist=16, m=8, iters=6)
    ix = oracle.OracleIndex(t["centroids"], t["codebook"], t["part_offsets"], t["codes"], t["row_ids"],
                            raw_vectors=t["raw"])
    q = (cent[rng.integers(0, 32, size=20)] + rng.normal(size=(20, 32))).astype(np.float32)
    truth, _, _, _ = oracle.flat_search(x, q, k=10)
    got, _, _, _ = ix.search(q, k=10, nprobe_min=8, nprobe_max=8, refine_factor=5)
    recall = np.mean([len(set(truth[i]) & set(got[i])) / 10 for i in range(20)])
    assert recall > 0.8


def test_merge_and_shard_plan(oracle):
    rng = np.random.default_rng(4)
    k, nq, nl = 5, 3, 4
    d = np.sort(rng.integers(0, 6, size=(nl, nq, k)).astype(np.float32), axis=2)
    ids = rng.permutation(nl * nq * k).reshape(nl, nq, k).astype(np.uint64)
    cnt = rng.integers(0, k + 1, size=(nl, nq)).astype(np.uint32)
    oi, od, oc = oracle.merge_topk(ids, d, cnt, k)
    for q in range(nq):
        pool = [(d[l, q, i], ids[l, q, i]) for l in range(nl) for i in range(cnt[l, q])]
        pool.sort()
        exp = pool[:k]
        assert oc[q] == len(exp)
        assert [(od[q, i], oi[q, i]) for i in range(len(exp))] == exp
    po = np.array([0, 10, 10, 40, 45, 100, 130], dtype=np.uint64)
    own = oracle.shard_plan(po, 2)
    lens = np.diff(po.astype(np.int64))
    loads = [lens[own == s].sum() for s in range(2)]
    # greedy LPT: 55->s0, 30->s1, 30->s1, 10->s0, 5->s0 (tie 65/60 -> lower load), 0->s1
    assert sorted(loads) == [65, 65] and own[4] == 0 and own[2] == 1 and own[5] == 1


def test_prefilter_postfilter_row_counts_like_the_reference(oracle):
    """rust/lancedb/src/query.rs:1759-1812 (test_execute): limit 10 with
    `only_if("id % 2 == 0")`: the prefilter (default) returns 10 rows, also with
    offset 1 (k = limit + offset rows are fetched, table/query.rs:231); the
    postfilter thins the unfiltered top 10, so fewer than 10 remain."""
    rng = np.random.default_rng(7)
    v = rng.random(size=(512, 4), dtype=np.float32)
    q = np.full((1, 4), 0.1, np.float32)
    even = np.arange(0, 512, 2, dtype=np.uint64)
    ids, dist, cnt, st = oracle.flat_search(v, q, k=10, allow_rowids=even)
    assert st == 0 and cnt[0] == 10 and (ids[0] % 2 == 0).all()
    ids11, _, cnt11, _ = oracle.flat_search(v, q, k=11, allow_rowids=even)  # limit 10, offset 1
    assert cnt11[0] == 11 and len(ids11[0][1:]) == 10
    plain, _, _, _ = oracle.flat_search(v, q, k=10)
    assert (plain[0] % 2 == 0).sum() < 10  # postfilter removes some of the 10
    # block list = complement of the allow list
    ids_b, dist_b, _, _ = oracle.flat_search(v, q, k=10, block_rowids=np.arange(1, 512, 2, dtype=np.uint64))
    assert (ids_b == ids).all() and (dist_b == dist).all()


@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
def test_encode_restatement_against_numpy_trainer(oracle, metric):
    """orc_ivfpq_encode (the checker of mi355_ivfpq_encode) against the independent
    numpy assignment / encoding of oracle/train.py (float64-free, different
    summation order: agreement up to rounding-level near-ties), plus the
    structural definitions: offsets = histogram prefix sums, order = stable sort."""
    rng = np.random.default_rng(8)
    cent = rng.normal(size=(24, 32)).astype(np.float32) * 2
    x = (cent[rng.integers(0, 24, size=6000)] + rng.normal(size=(6000, 32))).astype(np.float32)
    t = train.train_ivfpq(x, nlist=16, m=8, metric=metric, iters=4)
    po, codes, order, assign = oracle.ivfpq_encode(x, t["centroids"], t["codebook"], metric)
    assert (po[1:] - po[:-1] == np.bincount(assign, minlength=16)).all() and po[0] == 0
    assert (order == np.argsort(assign, kind="stable")).all()
    agree = assign[order] == t["assign"]  # the trainer's arrays are in index order
    assert agree.mean() > 0.999
    if agree.all():
        assert (po == t["part_offsets"]).all()
        assert (codes == t["codes"]).mean() > 0.999


@pytest.mark.parametrize("metric", ["l2", "cosine"])
def test_kmeans_restatement_against_numpy_lloyd(oracle, metric):
    """orc_kmeans_train against a plain numpy Lloyd iteration (float64 sums): same
    assignments on well-separated data, centroids equal to rounding."""
    rng = np.random.default_rng(5)
    cent = rng.normal(size=(10, 12)) * 6
    x = (cent[rng.integers(0, 10, size=3000)] + rng.normal(size=(3000, 12))).astype(np.float32)
    init = x[rng.choice(3000, size=10, replace=False)].copy()
    got, counts = oracle.kmeans_train(x, init, metric, 4)
    xs = x.astype(np.float64)
    if metric == "cosine":
        xs = xs / np.linalg.norm(xs, axis=1, keepdims=True)
    c = init.astype(np.float64)
    for _ in range(4):
        a = ((xs[:, None] - c[None]) ** 2).sum(-1).argmin(1)
        last = np.bincount(a, minlength=10)
        for p in range(10):
            if last[p]:
                c[p] = xs[a == p].mean(0)
    assert (counts == last).all()
    np.testing.assert_allclose(got, c, rtol=2e-4, atol=2e-5)
    r, a2 = oracle.ivf_residuals(x, got, metric)
    np.testing.assert_allclose(r, xs - got.astype(np.float64)[a2], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
@pytest.mark.parametrize("dim,nlist", [(768, 4096), (20, 13), (8, 8), (33, 70), (7, 3), (960, 122)])
def test_vectorised_coarse_is_the_scalar_chain_bit_for_bit(oracle, metric, dim, nlist):
    """orc_coarse_fast (AVX2: eight centroids per register, one d-ascending fmaf chain per lane — what orc_search and the
    bench's CPU baseline run) against orc_coarse (the scalar chain the contract is written in)."""
    from oracle import train
    rng = np.random.default_rng(dim * 31 + nlist)
    m = 1
    s = train.synthetic_index(200, dim, nlist, m, seed=dim)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], metric=metric)
    for _ in range(4):
        q = (rng.normal(size=dim) * rng.choice([1e-3, 1.0, 1e3])).astype(np.float32)
        a, b = o.coarse(q), o.coarse_fast(q)
        assert a.tobytes() == b.tobytes()
    q = rng.normal(size=dim).astype(np.float32)
    q[dim // 2] = np.nan
    a, b = o.coarse(q), o.coarse_fast(q)
    assert np.isnan(a).all() and np.isnan(b).all()
