"""Every num_sub_vectors the reference's builder produces runs the production scan.

`suggested_num_sub_vectors` (rust/lancedb/src/index/vector.rs:306-319) gives m = dim / 16 (else dim / 8, else 1):
256-d -> 16, 384-d -> 24, 960-d -> 60 (GIST1M, the dataset of the reference's published chart), 2048-d -> 128,
3072-d -> 192.  None of these is a kernel width of k_scan_skew (32 / 48 / 64 / 80 / 96 table columns); round 4 puts them
on it anyway (csrc/kernels_skew.h SkewShape): m <= 96 is padded with code 0 and an all-zero table column (`+ 0.0f` is
exact), m > 96 is scanned in slabs of <= 96 columns with the row's partial sum carried between slabs — the row sum stays
the contract's j-ascending chain, so every case here is compared with the oracle with `==`.

The generic kernel (k_scan_pair) stays the path of 4-bit codes and of MI355_INDEX_GENERIC_SCAN; the second half of the
file keeps it covered now that plain 8-bit indexes no longer reach it.
"""
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


def _expect_variant(m, nbits=8):
    """The layout mi355_index_open picks (csrc/ann_index_open.hip): the production scan unless its packed streams would be
    more than 8 x the source rows (padding to a 32-column tile, nibbles expanded to bytes): m = 1 .. 3 at 8 bits, m <= 6 at 4."""
    n_slabs = (m + 95) // 96
    per = (m + n_slabs - 1) // n_slabs
    M = max(32, (per + 15) // 16 * 16)
    return _abi.SCAN_SKEW if M * n_slabs <= 8 * ((m * nbits + 7) // 8) else _abi.SCAN_PAIR


def _pair(oracle, s, metric="l2", raw=None, **kw):
    g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s.get("row_ids"),
                               raw_vectors=raw, metric=metric, **kw)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s.get("row_ids"),
                           raw_vectors=raw, metric=metric)
    return g, o


# (m, dsub): the reference's default widths (dsub 16) plus a few that stress the shape rules: m = 1 (dim not divisible
# by 8), 100 -> two slabs of 64 with 28 padding columns, 112 -> 2 x 64 (16 padding), 144 -> 2 x 80, 240 -> 3 x 80,
# 288 -> 3 x 96 exactly, 300 -> 4 slabs.  Tables of 32 columns (m <= 32) run as two eight-wave workgroups per CU for k <= 64,
# and for k <= 128 while two sets of 192-row lists fit beside the residuals (m = 32 x 16 does not: 512-d residuals)
WIDTHS = [(8, 16), (16, 16), (24, 16), (32, 16), (32, 4), (60, 16), (128, 16), (192, 16), (1, 20), (2, 4), (33, 8), (100, 4), (112, 8),
          (144, 4), (240, 2), (288, 4), (300, 2)]


@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
@pytest.mark.parametrize("m,dsub", WIDTHS)
def test_every_builder_width_runs_the_production_scan(oracle, m, dsub, metric):
    dim, nlist = m * dsub, 12
    rng = np.random.default_rng(m * 7 + dsub)
    # partition lengths: empty, < 1 tile, exactly 16 / 32 tiles, ragged tails, > 32 tiles per partition
    lens = np.array([0, 1, 63, 64, 65, 1024, 1025, 3000, 0, 5000, 17, 2048], dtype=np.int64)
    n = int(lens.sum())
    s = train.synthetic_index(n, dim, nlist, m, seed=m)
    s["part_offsets"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    g, o = _pair(oracle, s, metric)
    q = (s["centroids"][rng.integers(0, nlist, size=7)] + rng.normal(0, 0.5, size=(7, dim))).astype(np.float32)
    for nprobe, k in ((1, 10), (5, 1), (12, 10), (12, 64), (12, 100), (12, 200), (12, 300)):
        _same(g.search(q, k=k, nprobe_min=nprobe, nprobe_max=nprobe), o.search(q, k=k, nprobe_min=nprobe, nprobe_max=nprobe))
    st = g.stats()
    assert st["scan_variant"] == _expect_variant(m)
    assert st["vectors_scanned"] == o.last_vectors_scanned and st["code_bytes_scanned"] == o.last_vectors_scanned * m
    # lance's per-partition transposed source layout packs to the same streams
    t = train.to_part_transposed(s["codes"], s["part_offsets"])
    g2 = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], t, s["row_ids"], metric=metric,
                                codes_layout=_abi.CODES_PART_TRANSPOSED)
    _same(g2.search(q, k=10, nprobe_min=12, nprobe_max=12), o.search(q, k=10, nprobe_min=12, nprobe_max=12))
    assert g2.stats()["scan_variant"] == _expect_variant(m)


@pytest.mark.parametrize("m,dim", [(24, 384), (60, 960), (128, 2048), (192, 3072)])
def test_reference_default_widths_batches_refine_filter_and_slices(oracle, m, dim):
    """The four default shapes end to end: a batch that fills every queue (stealing), ties, ranges, a NaN query, refine,
    maximum_nprobes, prefilter, k beyond one selection pass (the slabs are walked again per pass), and single queries cut
    into slices of tile positions (latency mode: a slice carries its partial sums through the slabs like a whole unit)."""
    rng = np.random.default_rng(m)
    n, nlist = 90_000, 20
    s = train.synthetic_index(n, dim, nlist, m, seed=m + 1, skew=1.0, empty_parts=2)
    raw = rng.normal(size=(n, dim)).astype(np.float32)
    g, o = _pair(oracle, s, raw=raw)
    g.configure(graph=False, coalesce=False)
    qb = (s["centroids"][rng.integers(0, nlist, size=300)] + rng.normal(0, 0.5, size=(300, dim))).astype(np.float32)
    exp = o.search(qb, k=10, nprobe_min=6, nprobe_max=6)
    _same(g.search(qb, k=10, nprobe_min=6, nprobe_max=6), exp)
    assert g.stats()["scan_variant"] == _abi.SCAN_SKEW
    lo, hi = float(exp[1][0, 2]), float(exp[1][0, 8])
    kw = dict(k=10, nprobe_min=6, nprobe_max=6, lower_bound=lo, upper_bound=hi)
    _same(g.search(qb[:40], **kw), o.search(qb[:40], **kw))
    qn = qb[:4].copy()
    qn[1, 5] = np.nan
    _same(g.search(qn, k=5, nprobe_min=2, nprobe_max=2), o.search(qn, k=5, nprobe_min=2, nprobe_max=2))
    allow = np.sort(rng.choice(n, size=n // 3, replace=False).astype(np.uint64))
    for kw in (dict(k=10, nprobe_min=4, nprobe_max=4, refine_factor=10), dict(k=70, nprobe_min=2, nprobe_max=nlist),
               dict(k=10, nprobe_min=5, nprobe_max=5, allow_rowids=allow), dict(k=600, nprobe_min=3, nprobe_max=3),
               dict(k=10, nprobe_min=4, nprobe_max=4, refine_factor=30)):
        _same(g.search(qb[:16], **kw), o.search(qb[:16], **kw))
    for nq, nprobe in ((1, 8), (1, 20), (3, 12)):
        q = qb[100:100 + nq]
        for kw in (dict(k=10), dict(k=100), dict(k=250), dict(k=10, refine_factor=10)):
            kw = dict(nprobe_min=nprobe, nprobe_max=nprobe, **kw)
            _same(g.search(q, **kw), o.search(q, **kw))
    # every row of the index shares one code: all distances of a partition tie, the row id alone orders them
    s["codes"][:] = s["codes"][0]
    g3, o3 = _pair(oracle, s)
    _same(g3.search(qb[:8], k=25, nprobe_min=3, nprobe_max=3), o3.search(qb[:8], k=25, nprobe_min=3, nprobe_max=3))


def test_padded_and_slabbed_shards_and_local_arrays(oracle):
    """A shard handle cut from the global arrays and one opened from its own partitions only pack the same slabs."""
    m, dim, nlist, n = 128, 512, 16, 40_000
    s = train.synthetic_index(n, dim, nlist, m, seed=3, skew=0.6)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    q = np.random.default_rng(0).normal(size=(9, dim)).astype(np.float32)
    owner = lancedb_amd.shard_plan(s["part_offsets"], 2)
    parts = []
    for r in range(2):
        sh = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"],
                                    shard_count=2, shard_rank=r)
        got = sh.search(q, k=10, nprobe_min=6, nprobe_max=6)
        assert sh.stats()["scan_variant"] == _abi.SCAN_SKEW
        # the same shard from local arrays
        po = s["part_offsets"].astype(np.int64)
        rows = np.concatenate([np.arange(po[p], po[p + 1]) for p in range(nlist) if owner[p] == r] or [np.zeros(0, np.int64)])
        sl = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"][rows], s["row_ids"][rows],
                                    shard_count=2, shard_rank=r, local_arrays=True)
        got2 = sl.search(q, k=10, nprobe_min=6, nprobe_max=6)
        assert (got.rowids == got2.rowids).all() and (got.distances == got2.distances).all() and (got.counts == got2.counts).all()
        parts.append(got)
    # the union of the two shards' lists, merged, is the unsharded result
    DA = lancedb_amd.DeviceArray
    mi, md, mc = lancedb_amd.merge_topk(DA.from_numpy(np.stack([p.rowids.astype(np.int64) for p in parts])),
                                        DA.from_numpy(np.stack([p.distances for p in parts])),
                                        DA.from_numpy(np.stack([p.counts.astype(np.int32) for p in parts])), 10)
    e_ids, e_dist, e_cnt, _ = o.search(q, k=10, nprobe_min=6, nprobe_max=6)
    assert (mi.numpy().astype(np.uint64) == e_ids).all() and (md.numpy() == e_dist).all()
    assert (mc.numpy().astype(np.uint32) == e_cnt).all()


# ------------------------------------------------------------------ the generic kernel stays covered ----
@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
@pytest.mark.parametrize("shape", [(5000, 32, 16, 8), (20000, 64, 64, 16), (3000, 24, 7, 3), (30000, 192, 8, 48)])
def test_generic_scan_flag_matches_oracle(oracle, metric, shape):
    n, dim, nlist, m = shape
    s = train.synthetic_index(n, dim, nlist, m, seed=n + m, empty_parts=min(2, nlist // 4))
    rng = np.random.default_rng(3)
    raw = rng.normal(size=(n, dim)).astype(np.float32)
    g, o = _pair(oracle, s, metric, raw=raw, generic_scan=True)
    q = rng.normal(size=(9, dim)).astype(np.float32)
    for nprobe in (1, max(1, nlist // 4), nlist):
        for k in (1, 10, 70, 300):
            _same(g.search(q, k=k, nprobe_min=nprobe, nprobe_max=nprobe), o.search(q, k=k, nprobe_min=nprobe, nprobe_max=nprobe))
    assert g.stats()["scan_variant"] == _abi.SCAN_PAIR
    exp = o.search(q, k=20, nprobe_min=nlist, nprobe_max=nlist)
    kw = dict(k=20, nprobe_min=nlist, nprobe_max=nlist, lower_bound=float(exp[1][0, 3]), upper_bound=float(exp[1][0, 12]))
    _same(g.search(q, **kw), o.search(q, **kw))
    allow = np.sort(rng.choice(n, size=n // 3, replace=False).astype(np.uint64))
    for kw in (dict(k=10, nprobe_min=4, nprobe_max=4, refine_factor=5), dict(k=30, nprobe_min=1, nprobe_max=nlist),
               dict(k=10, nprobe_min=3, nprobe_max=3, allow_rowids=allow)):
        _same(g.search(q, **kw), o.search(q, **kw))
    qb = rng.normal(size=(300, dim)).astype(np.float32)
    _same(g.search(qb, k=10, nprobe_min=3, nprobe_max=3), o.search(qb, k=10, nprobe_min=3, nprobe_max=3))


def test_multi_slab_scan_of_a_very_long_partition(oracle):
    """m > 96 keeps 8 B of partial sums per (tile position, unit, lane) of the LONGEST partition per workgroup; a partition of
    2.2 M rows makes that 8.8 MB per workgroup, more than the 2 GiB the handle allows for 256 of them: the launch then runs on
    fewer workgroups (the persistent kernel takes any count).  Sliced (40 queries x 1 probe = 320 work items) and a batch."""
    rng = np.random.default_rng(11)
    m, dsub, nlist = 128, 2, 2
    dim = m * dsub
    lens = np.array([2_200_000, 3_000], dtype=np.int64)
    n = int(lens.sum())
    s = train.synthetic_index(n, dim, nlist, m, seed=5)
    s["part_offsets"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    s["centroids"][1] += 50.0  # every query probes the long partition first
    g, o = _pair(oracle, s)
    q = (s["centroids"][0] + rng.normal(0, 0.5, size=(40, dim))).astype(np.float32)
    for kw in (dict(k=10, nprobe_min=1, nprobe_max=1), dict(k=100, nprobe_min=2, nprobe_max=2)):
        _same(g.search(q, **kw), o.search(q, **kw))
    assert g.stats()["scan_variant"] == _abi.SCAN_SKEW


@pytest.mark.parametrize("seed", range(12))
def test_random_shapes_against_the_oracle(oracle, seed):
    """Seeded random shapes across the SkewShape rules (padding, 2-8 slabs), sub-vector lengths with and without the 16-B table
    builder, all metrics, k across the list-length and multi-pass regimes, ranges, sliced and chip-filling batches."""
    rng = np.random.default_rng(1000 + seed)
    m = int(rng.choice([1, 3, 7, 8, 12, 20, 24, 31, 33, 40, 60, 72, 90, 97, 110, 128, 150, 192, 200, 256, 320, 400, 500, 700]))
    dsub = int(rng.choice([1, 2, 3, 4, 8, 16] if m <= 128 else [1, 2, 4]))
    dim = m * dsub
    nlist = int(rng.integers(3, 20))
    lens = rng.integers(0, 6000, size=nlist)
    lens[rng.integers(0, nlist)] = int(rng.integers(6000, 40000))
    n = int(lens.sum())
    metric = ["l2", "cosine", "dot"][seed % 3]
    s = train.synthetic_index(n, dim, nlist, m, seed=seed)
    s["part_offsets"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    g, o = _pair(oracle, s, metric)
    for nq in (1, 5, int(rng.integers(20, 400))):
        q = (s["centroids"][rng.integers(0, nlist, size=nq)] + rng.normal(0, 0.5, size=(nq, dim))).astype(np.float32)
        k = int(rng.choice([1, 10, 64, 100, 129, 256, 300]))
        nprobe = int(rng.integers(1, nlist + 1))
        exp = o.search(q, k=k, nprobe_min=nprobe, nprobe_max=nprobe)
        _same(g.search(q, k=k, nprobe_min=nprobe, nprobe_max=nprobe), exp)
        fin = exp[1][0][np.isfinite(exp[1][0])]
        if len(fin) > 4:
            kw = dict(k=k, nprobe_min=nprobe, nprobe_max=nprobe, lower_bound=float(fin[1]), upper_bound=float(fin[-2]))
            _same(g.search(q, **kw), o.search(q, **kw))
    assert g.stats()["scan_variant"] == _expect_variant(m)
