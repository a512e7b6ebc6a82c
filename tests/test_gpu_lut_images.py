"""Batch-level distance tables (csrc/kernels_lut.h) against in-item table builds and against the oracle.

An index whose sub-vectors are 16 floats long — `num_sub_vectors = dim / 16`, what the reference's builder picks by
default (rust/lancedb/src/index/vector.rs:306-319) — gets the PQ distance tables of a whole batch (SURVEY.md section 8a row
a14: one table per (query, probed partition)) from `k_lut_images`, which keeps the codebook in registers; the scan work
items copy their table image instead of streaming 256 * dim * 4 bytes of codebook each.  Same operations in the same
order, so: images == in-item builds == oracle, compared with `==` (ids AND distances).
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


def _equal(a, b):
    assert (a.counts == b.counts).all() and (a.rowids == b.rowids).all() and (a.distances == b.distances).all()


# (m, dim): 48 x 16 = the reference's default for 768-d (a plain kernel width); 24 -> padded to 32 columns, two workgroups
# per CU; 96 x 16 = 1536-d (C5's shape); 60 -> padded to 64 (GIST's 960-d); 112 -> two slabs of 64 (8 padding columns
# each... 56 real); 192 -> two slabs of 96 (3072-d); 9 -> an odd m (the last column pair of a residual row is half padding)
SHAPES = [(48, 768), (24, 384), (96, 1536), (60, 960), (112, 1792), (192, 3072), (9, 144)]


@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
@pytest.mark.parametrize("m,dim", SHAPES)
def test_images_equal_inline_builds_and_the_oracle(oracle, m, dim, metric):
    rng = np.random.default_rng(m)
    nlist = 14
    lens = np.array([0, 1, 63, 64, 65, 1024, 1025, 3000, 0, 5000, 17, 2048, 8192, 700], dtype=np.int64)
    n = int(lens.sum())
    s = train.synthetic_index(n, dim, nlist, m, seed=m + 3)
    s["part_offsets"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], metric=metric)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], metric=metric)
    # (520 queries: batches of up to 512 (query, partition) pairs that cannot fill the chip are cut by rows and build their tables
    #  in the work items — round 6, tests/test_gpu_latency_mode.py; with one probe these 520 pairs are cut in two slices: image + slices)
    q = (s["centroids"][rng.integers(0, nlist, size=520)] + rng.normal(0, 0.5, size=(520, dim))).astype(np.float32)
    for nprobe, k in ((1, 10), (5, 1), (14, 10), (14, 64), (14, 100), (14, 128)):
        kw = dict(k=k, nprobe_min=nprobe, nprobe_max=nprobe)
        g.configure(lut_inline=False)
        a = g.search(q, **kw)
        assert g.stats()["lut_images"] == 1, "the batch-level table kernel did not run"
        g.configure(lut_inline=True)
        b = g.search(q, **kw)
        assert g.stats()["lut_images"] == 0
        _equal(a, b)
        _same(a, o.search(q, **kw))
    # candidate lists beyond the image kernels' instantiations keep the in-item build
    g.configure(lut_inline=False)
    _same(g.search(q[:5], k=300, nprobe_min=6, nprobe_max=6), o.search(q[:5], k=300, nprobe_min=6, nprobe_max=6))
    assert g.stats()["lut_images"] == 0


def test_images_with_slices_refine_filter_ranges_and_second_pass(oracle):
    """The default 768-d shape end to end: single queries and small batches (cut by rows; they build their tables in the work
    items — round 6), batches cut into a fixed number of slices (every slice of a pair copies the SAME image), refine, prefilter,
    distance ranges, a NaN query, and maximum_nprobes (the second pass runs behind a device-side batch size and builds its tables
    in the items)."""
    m, dim, n, nlist = 48, 768, 120_000, 16
    rng = np.random.default_rng(5)
    s = train.synthetic_index(n, dim, nlist, m, seed=11, skew=1.0, empty_parts=2)
    raw = rng.normal(size=(n, dim)).astype(np.float32)
    g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw)
    g.configure(graph=False, coalesce=False)
    qb = (s["centroids"][rng.integers(0, nlist, size=200)] + rng.normal(0, 0.5, size=(200, dim))).astype(np.float32)
    for nq, nprobe in ((1, 8), (1, 16), (3, 12), (8, 5), (100, 6), (200, 6)):
        q = qb[:nq]
        for kw in (dict(k=10), dict(k=100), dict(k=10, refine_factor=10)):
            kw = dict(nprobe_min=nprobe, nprobe_max=nprobe, **kw)
            _same(g.search(q, **kw), o.search(q, **kw))
            # up to 512 pairs: cut by rows, tables built in the items; 600 pairs: two slices per pair + images; 1200: whole partitions
            assert g.stats()["lut_images"] == (1 if nq * nprobe > 512 else 0)
    exp = o.search(qb, k=10, nprobe_min=6, nprobe_max=6)
    lo, hi = float(exp[1][0, 2]), float(exp[1][0, 8])
    kw = dict(k=10, nprobe_min=6, nprobe_max=6, lower_bound=lo, upper_bound=hi)
    _same(g.search(qb[:40], **kw), o.search(qb[:40], **kw))
    qn = qb[:4].copy()
    qn[1, 5] = np.nan
    _same(g.search(qn, k=5, nprobe_min=2, nprobe_max=2), o.search(qn, k=5, nprobe_min=2, nprobe_max=2))
    allow = np.sort(rng.choice(n, size=n // 3, replace=False).astype(np.uint64))
    kw = dict(k=10, nprobe_min=5, nprobe_max=5, allow_rowids=allow)
    _same(g.search(qb[:16], **kw), o.search(qb[:16], **kw))
    kw = dict(k=70, nprobe_min=2, nprobe_max=nlist)  # (k = 70 rows are not found in two partitions by every query: second pass)
    _same(g.search(qb[:16], **kw), o.search(qb[:16], **kw))
    # device I/O of the same call
    dq = lancedb_amd.DeviceArray.from_numpy(qb)
    r = g.search(dq, k=10, nprobe_min=6, nprobe_max=6)
    g.sync()
    assert (r.rowids.numpy().view(np.uint64) == exp[0]).all() and (r.distances.numpy() == exp[1]).all()


def test_images_on_a_sharded_handle_and_external_probes(oracle):
    """A shard handle makes work items for the partitions it owns only: the table kernel walks the planner's work list, so
    the pairs of foreign partitions cost nothing and leave no image.  External probe lists (mi355_search_probes) may name
    partitions outside the index."""
    m, dim, n, nlist = 48, 768, 60_000, 24
    rng = np.random.default_rng(9)
    s = train.synthetic_index(n, dim, nlist, m, seed=21, skew=0.7)
    q = (s["centroids"][rng.integers(0, nlist, size=80)] + rng.normal(0, 0.5, size=(80, dim))).astype(np.float32)  # (640 pairs: images)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    full = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    exp = o.search(q, k=10, nprobe_min=8, nprobe_max=8)
    _same(full.search(q, k=10, nprobe_min=8, nprobe_max=8), exp)
    shards = [lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], shard_count=3, shard_rank=r)
              for r in range(3)]
    parts = []
    for sh in shards:
        a = sh.search(q, k=10, nprobe_min=8, nprobe_max=8)
        assert sh.stats()["lut_images"] == 1
        sh.configure(lut_inline=True)
        b = sh.search(q, k=10, nprobe_min=8, nprobe_max=8)
        _equal(a, b)
        parts.append(a)
    for b in range(len(q)):  # the k-way merge of the shards' lists in the (distance, rowid) order = the unsharded result
        rows = [(float(p.distances[b, j]), int(p.rowids[b, j])) for p in parts for j in range(int(p.counts[b]))]
        rows.sort()
        top = rows[:10]
        assert [r[1] for r in top] == [int(x) for x in exp[0][b, :len(top)]] and len(top) == int(exp[2][b])
        assert [np.float32(r[0]) for r in top] == list(exp[1][b, :len(top)])
    probes = np.tile(np.array([3, 7, 7, nlist + 5, 0], dtype=np.uint64), (4, 1))  # a duplicate and an id outside the index
    with pytest.raises(lancedb_amd.InvalidInput):
        full.search_probes(q[:4], probes, k=10)
    probes = np.tile(np.array([3, 7, 11, 0], dtype=np.uint64), (4, 1))
    a = full.search_probes(q[:4], probes, k=10)
    full.configure(lut_inline=True)
    b = full.search_probes(q[:4], probes, k=10)
    _equal(a, b)
