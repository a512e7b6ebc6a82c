"""Latency mode of the production scan: a batch too small to give every CU a (query, partition) work item
is cut into slices of tile positions (SkewArgs::n_slices, csrc/kernels_skew.h) — one query then runs on
all 256 CUs instead of on `nprobe` of them.  Results must not depend on the slicing: every case is the
unsliced oracle result, bit for bit (callers are single-query tokio workers,
python/src/runtime.rs:31-37)."""
import numpy as np
import pytest

import lancedb_amd
from oracle import train

pytestmark = pytest.mark.gpu


def _same(got, exp):
    ids, dist, cnt, st = exp
    assert st == 0
    assert (got.counts == cnt).all() and (got.rowids == ids).all() and (got.distances == dist).all()


@pytest.mark.parametrize("m,dim", [(32, 128), (96, 768)])
def test_sliced_work_items_return_the_unsliced_result(oracle, m, dim):
    rng = np.random.default_rng(m)
    n, nlist = 600_000, 40
    # strong skew: partitions from a few hundred rows (streams with 0-2 tiles: empty slices) to ~80 k rows
    s = train.synthetic_index(n, dim, nlist, m, seed=4, skew=1.2, empty_parts=2)
    raw = rng.normal(size=(n, dim)).astype(np.float32)
    ix = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw)
    ix.configure(graph=False, coalesce=False)
    allow = np.sort(rng.choice(n, size=n // 3, replace=False).astype(np.uint64))
    for nq, nprobe in ((1, 8), (1, 32), (3, 12), (2, 40)):
        q = (s["centroids"][rng.integers(0, nlist, size=nq)] + rng.normal(0, 0.5, size=(nq, dim))).astype(np.float32)
        ub = float(o.search(q, k=40, nprobe_min=nprobe, nprobe_max=nprobe)[1][0, 30])
        for kw in (dict(k=10), dict(k=100), dict(k=250), dict(k=10, refine_factor=10), dict(k=25, upper_bound=ub),
                   dict(k=10, allow_rowids=allow)):
            kw = dict(nprobe_min=nprobe, nprobe_max=nprobe, **kw)
            _same(ix.search(q, **kw), o.search(q, **kw))
        st = ix.stats()
        pairs = nq * nprobe
        # (round 6) the planner cuts these batches by rows on the device; the host counts the candidate slots it laid out:
        # the most a pair may be cut into — 16, or what keeps a query's slots within the block merge's 1024 sources
        slices = min(16, 1024 // nprobe)
        assert st["work_items"] == pairs * slices, (st["work_items"], pairs, slices)
    # a batch that fills the chip keeps whole partitions
    q = rng.normal(size=(128, dim)).astype(np.float32)
    _same(ix.search(q, k=10, nprobe_min=8, nprobe_max=8), o.search(q, k=10, nprobe_min=8, nprobe_max=8))
    assert ix.stats()["work_items"] == 128 * 8
    # in between (512 pairs on 256 CUs): two slices per pair, so that the longest partition does not decide alone
    _same(ix.search(q[:64], k=10, nprobe_min=8, nprobe_max=8), o.search(q[:64], k=10, nprobe_min=8, nprobe_max=8))
    assert ix.stats()["work_items"] == 64 * 8 * 16  # (slots: the sparse planner cuts by rows, up to 16 items per pair)


@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
@pytest.mark.parametrize("dim,m", [(128, 32), (50, 10)])
def test_small_batches_take_the_fused_prep_and_coarse_launch(oracle, metric, dim, m):
    """<= 8 queries: prep + coarse run as one launch of single-wave workgroups (k_coarse_small); the probe lists and
    the results must be the oracle's for every metric, also for a dimension that is not a multiple of 4."""
    rng = np.random.default_rng(dim)
    s = train.synthetic_index(30000, dim, 48, m, seed=9, skew=0.8, empty_parts=1)
    ix = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], metric=metric)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], metric=metric)
    for nq in (1, 2, 7, 8, 9):
        q = (s["centroids"][rng.integers(0, 48, size=nq)] + rng.normal(0, 0.4, size=(nq, dim))).astype(np.float32)
        _same(ix.search(q, k=10, nprobe_min=6, nprobe_max=6), o.search(q, k=10, nprobe_min=6, nprobe_max=6))
        # maximum_nprobes expansion on top of the small path (the second pass runs behind a device-side mask)
        ub = float(o.search(q, k=30, nprobe_min=20, nprobe_max=20)[1][0, 20])
        kw = dict(k=25, nprobe_min=2, nprobe_max=20, upper_bound=ub)
        _same(ix.search(q, **kw), o.search(q, **kw))


@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
@pytest.mark.parametrize("nlist,dim,m", [(5000, 768, 96), (9000, 96, 12), (2500, 200, 25)])
def test_small_batches_split_centroid_rows_across_lanes(oracle, metric, nlist, dim, m):
    """Centroid tables that would leave most CUs without a wave at one centroid per lane give 4 / 8 / 16 lanes to a
    centroid (k_coarse_split): rows staged in rounds of 64 / 128 / 256 pieces, the last round partial, queries dealt
    to the lanes of a centroid.  Probe lists and results are the oracle's."""
    rng = np.random.default_rng(nlist)
    s = train.synthetic_index(8 * nlist, dim, nlist, m, seed=5, skew=0.5, empty_parts=3)
    ix = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], metric=metric)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], metric=metric)
    ix.configure(graph=False, coalesce=False)
    for nq in (1, 3, 5, 8):
        q = (s["centroids"][rng.integers(0, nlist, size=nq)] + rng.normal(0, 0.4, size=(nq, dim))).astype(np.float32)
        _same(ix.search(q, k=10, nprobe_min=24, nprobe_max=24), o.search(q, k=10, nprobe_min=24, nprobe_max=24))


@pytest.mark.parametrize("dup", [3000, 40, 1])
def test_block_merge_short_list_with_ties_and_overflow(oracle, dup):
    """A handful of queries reduce their work items' slots with a 16-wave block that first cuts them to a short list
    under a bound (k_merge_cands<KPL, 16>).  Rows that share a code vector tie exactly: `dup` = 3000 makes every slot
    of every slice tie with thousands of others (the short list overflows and one wave walks all slots), 40 ties the
    winners across slices at the bound, 1 is the plain case.  Any k up to 1024 takes the block; results are the
    oracle's (ties by row id)."""
    rng = np.random.default_rng(dup)
    n, dim, m, nlist = 400_000, 64, 16, 24
    s = train.synthetic_index(n, dim, nlist, m, seed=11, skew=0.6, empty_parts=1)
    if dup > 1:
        base = rng.integers(0, 256, size=((n + dup - 1) // dup, m), dtype=np.uint8)
        s["codes"] = np.ascontiguousarray(base[rng.integers(0, base.shape[0], size=n)])
    ix = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    ix.configure(graph=False, coalesce=False)
    for nq, nprobe in ((1, 8), (2, 20), (1, 24)):
        q = (s["centroids"][rng.integers(0, nlist, size=nq)] + rng.normal(0, 0.5, size=(nq, dim))).astype(np.float32)
        for k in (1, 10, 64, 100, 250, 1000, 1500):
            kw = dict(k=k, nprobe_min=nprobe, nprobe_max=nprobe)
            _same(ix.search(q, **kw), o.search(q, **kw))


@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_latency_front_probe_selection_with_tied_and_degenerate_scores(oracle, metric):
    """<= 8 queries on the production scan (L2 / dot, dim % 4 == 0, nlist <= 8192): k_coarse_lat writes the raw dot chains,
    k_select_plan finishes the scores, selects the probes over the key bits that differ and lays out the work list.
    Duplicated centroids tie exactly at the selection threshold (ties go to the lowest partition id), identical
    centroids leave no differing bit at all, a far-away query mixes signs (dot) — the probe sets, and therefore the
    results, must be the oracle's; nprobe = 1, = nlist and in between."""
    rng = np.random.default_rng(3)
    n, dim, m, nlist = 60_000, 64, 16, 96
    s = train.synthetic_index(n, dim, nlist, m, seed=21, skew=0.7, empty_parts=2)
    cen = s["centroids"].copy()
    cen[10:40] = cen[5]          # 31 partitions at exactly the same score for every query
    cen[60:64] = cen[59]
    for variant in ("dups", "all_equal"):
        c = cen if variant == "dups" else np.repeat(cen[:1], nlist, axis=0)
        ix = lancedb_amd.IvfPqIndex(c, s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], metric=metric)
        o = oracle.OracleIndex(c, s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], metric=metric)
        ix.configure(graph=False, coalesce=False)
        for nq in (1, 2, 5, 8):
            q = (c[rng.integers(0, nlist, size=nq)] + rng.normal(0, 0.3, size=(nq, dim))).astype(np.float32)
            q[0] = c[5]  # sits on the duplicated centroid: the tie is at the TOP of the list
            if nq > 1:
                q[1] = -40.0 * c[7]  # far away, opposite side: dot scores of both signs
            for nprobe in (1, 7, 20, 33, 64, nlist):
                kw = dict(k=10, nprobe_min=nprobe, nprobe_max=nprobe)
                _same(ix.search(q, **kw), o.search(q, **kw))


def test_latency_front_serves_concurrent_shapes_back_to_back(oracle):
    """The ticket word of k_select_plan returns to zero after every launch: alternating batch sizes (1, 8, 3, 9 = the
    general front, 1 ...) on one handle keep returning the oracle's results."""
    rng = np.random.default_rng(8)
    n, dim, m, nlist = 200_000, 128, 32, 256
    s = train.synthetic_index(n, dim, nlist, m, seed=2, skew=0.5)
    ix = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    for nq in (1, 8, 3, 9, 1, 64, 2, 1, 8):
        q = (s["centroids"][rng.integers(0, nlist, size=nq)] + rng.normal(0, 0.5, size=(nq, dim))).astype(np.float32)
        kw = dict(k=10, nprobe_min=32, nprobe_max=32)
        _same(ix.search(q, **kw), o.search(q, **kw))


@pytest.mark.parametrize("seed", range(16))
def test_latency_front_random_shapes_against_the_oracle(oracle, seed):
    """Seeded random shapes across the latency front's branches: 4 / 8 / 16 lanes per centroid (nlist 8192 / ~4096 / small), centroid
    counts that do not fill the last workgroup, one to eight queries (the single host query rides in the kernel's argument block when
    dim <= 896, otherwise through the staging copy), nprobe from 1 to nlist (more than 512 pairs falls back to the general planner),
    all three metrics, refine and ranges — every result is the oracle's, bit for bit."""
    rng = np.random.default_rng(7000 + seed)
    nlist = int(rng.choice([2, 3, 17, 64, 100, 511, 1024, 2049, 4096, 8192]))
    dsub = int(rng.choice([1, 2, 4, 8, 16]))
    m = int(rng.choice([4, 8, 12, 24, 32, 48, 96]))
    dim = m * dsub
    if dim % 4:
        dim, dsub = m * 4, 4
    if dim > 1536:
        m, dim = 96, 96 * dsub
    metric = ["l2", "cosine", "dot"][seed % 3]
    n = int(max(3000, min(60_000, nlist * int(rng.integers(2, 12)))))
    s = train.synthetic_index(n, dim, nlist, m, seed=seed, skew=0.7, empty_parts=min(2, nlist // 4))
    raw = rng.normal(size=(n, dim)).astype(np.float32)
    g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw, metric=metric)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw, metric=metric)
    g.configure(graph=False, coalesce=False)
    for nq in (1, int(rng.integers(2, 9)), 8):
        q = (s["centroids"][rng.integers(0, nlist, size=nq)] + rng.normal(0, 0.4, size=(nq, dim))).astype(np.float32)
        for nprobe in sorted({1, min(nlist, int(rng.integers(1, 80))), min(nlist, 64), nlist if nlist <= 128 else 100}):
            k = int(rng.choice([1, 10, 33]))
            kw = dict(k=k, nprobe_min=nprobe, nprobe_max=nprobe)
            _same(g.search(q, **kw), o.search(q, **kw))
        kw = dict(k=10, nprobe_min=min(nlist, 8), nprobe_max=min(nlist, 8), refine_factor=5)
        _same(g.search(q, **kw), o.search(q, **kw))
