"""The world > 1 exchange of csrc/ann_comm.hip, executed on ONE GPU through the loopback
transport (mi355_comm_create_loopback): `world` ranks of this process, one thread and one shard
handle each; the gather is a host rendezvous + device copies into the same slab layout
ncclAllGather fills, so sharded_ann / sharded_finish / the collective maximum_nprobes pass / the
sharded coarse stage / the flat row-sharded search are the code an 8-GPU node runs.  Every rank's
result is compared with the UNSHARDED oracle (row ids and distances with ==).

SURVEY.md section 8e is the contract (the reference has no collective: section 2, section 5 last row)."""
import os
import sys
import threading

import numpy as np
import pytest

import lancedb_amd
from lancedb_amd import _abi
from lancedb_amd.distributed import Comm, ShardedFlatSearcher, ShardedSearcher, run_ranks
from oracle import train

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
DA = lancedb_amd.DeviceArray


def _same(got, exp, what=""):
    ids, dist, cnt, st = exp
    assert st == 0
    assert (np.asarray(got.counts) == cnt).all(), what
    assert (np.asarray(got.rowids) == ids).all(), what
    assert (np.asarray(got.distances) == dist).all(), what


def _shards(s, raw, world, **kw):
    return [lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"],
                                   raw_vectors=raw, shard_count=world, shard_rank=r, **kw) for r in range(world)]


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("m,dim", [(8, 32), (32, 128)])
def test_every_rank_of_a_loopback_world_returns_the_unsharded_result(oracle, world, m, dim):
    rng = np.random.default_rng(world * 100 + m)
    s = train.synthetic_index(40000, dim, 48, m, seed=13 + world, skew=0.9, empty_parts=2)
    raw = rng.normal(size=(40000, dim)).astype(np.float32)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw)
    q = rng.normal(size=(33, dim)).astype(np.float32)
    ub = float(o.search(q, k=30, nprobe_min=11, nprobe_max=11)[1][0, 25])
    cases = [dict(k=10, nprobe_min=16, nprobe_max=16),
             dict(k=10, nprobe_min=8, nprobe_max=8, refine_factor=10),   # owner-side refine, second gather
             dict(k=7, nprobe_min=5, nprobe_max=5, refine_factor=50),
             dict(k=300, nprobe_min=6, nprobe_max=6),                    # lists longer than one selection pass
             dict(k=20, nprobe_min=2, nprobe_max=11, upper_bound=ub),    # collective maximum_nprobes second pass
             dict(k=20, nprobe_min=2, nprobe_max=11, upper_bound=ub, refine_factor=3)]
    exp = [o.search(q, **kw) for kw in cases]
    shards = _shards(s, raw, world)
    comms = Comm.loopback(world)
    for coarse in (False, True):
        def rank_fn(r):
            sh = ShardedSearcher(shards[r], comms[r], shard_coarse=coarse)
            return [sh.search(q, _abi.make_params(**kw)) for kw in cases]
        got = run_ranks([lambda r=r: rank_fn(r) for r in range(world)])
        for r in range(world):
            for kw, g, e in zip(cases, got[r], exp):
                _same(g, e, f"world {world} rank {r} coarse {coarse} {kw}")
    st = [c.stats() for c in comms]
    total = sum(st[0]["rows_scanned"])
    assert total > 0 and all(x["rows_scanned"] == st[0]["rows_scanned"] for x in st)  # identical on every rank
    assert st[0]["world"] == world and st[0]["imbalance"] >= 1.0 and st[0]["n_gathers"] >= 2


def test_overlapped_device_calls_pipeline_across_batches(oracle):
    """Device-I/O calls queue their exchange on the communicator's stream: several calls in flight
    (double-buffered slabs), outputs complete after sync(); results == unsharded oracle for every batch."""
    world, dim, m = 4, 128, 32
    rng = np.random.default_rng(77)
    s = train.synthetic_index(60000, dim, 64, m, seed=3, skew=0.9, empty_parts=1)
    raw = rng.normal(size=(60000, dim)).astype(np.float32)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw)
    shards = _shards(s, raw, world)
    comms = Comm.loopback(world)
    n_batches = 5
    qs = [rng.normal(size=(24, dim)).astype(np.float32) for _ in range(n_batches)]
    for kw in (dict(k=10, nprobe_min=12, nprobe_max=12), dict(k=10, nprobe_min=12, nprobe_max=12, refine_factor=5)):
        exp = [o.search(q, **kw) for q in qs]

        def rank_fn(r):
            sh = ShardedSearcher(shards[r], comms[r])
            dq = [DA.from_numpy(q) for q in qs]
            outs = [(DA((24, 10), np.int64), DA((24, 10), np.float32), DA((24,), np.int32)) for _ in qs]
            res = [sh.search(dq[i], _abi.make_params(**kw), out=outs[i]) for i in range(n_batches)]  # no sync in between
            shards[r].sync()
            return [(x.rowids.numpy().view(np.uint64), x.distances.numpy(), x.counts.numpy()) for x in res], comms[r].stats()
        got = run_ranks([lambda r=r: rank_fn(r) for r in range(world)])
        for r in range(world):
            res, st = got[r]
            assert st["overlapped"] and st["us_exchange"] > 0
            for (ids, dist, cnt), e in zip(res, exp):
                assert (cnt == e[2]).all() and (ids == e[0]).all() and (dist == e[1]).all(), (r, kw)
    # the unsharded entry points of a shard handle join a pending exchange before they use the stream
    r0 = shards[0].search(qs[0], k=5, nprobe_min=4, nprobe_max=4)
    assert r0.counts.shape == (24,)


def test_sharded_coarse_at_nlist_65536(oracle):
    """C4's shape: the centroid matrix itself is sharded (MI355_SHARD_COARSE): every rank scores 1/world of the
    65536 centroids, one gather of nprobe (partition, distance) pairs per query per rank selects the probe list."""
    world, dim, m, nlist = 8, 32, 8, 65536
    rng = np.random.default_rng(5)
    s = train.synthetic_index(150000, dim, nlist, m, seed=21, skew=0.7)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    q = (s["centroids"][rng.integers(0, nlist, size=19)] + rng.normal(0, 0.3, size=(19, dim))).astype(np.float32)
    kw = dict(k=10, nprobe_min=128, nprobe_max=128)
    exp = o.search(q, **kw)
    shards = _shards(s, None, world)
    comms = Comm.loopback(world)
    got = run_ranks([lambda r=r: ShardedSearcher(shards[r], comms[r], shard_coarse=True).search(q, _abi.make_params(**kw))
                     for r in range(world)])
    for r in range(world):
        _same(got[r], exp, f"rank {r}")


def test_rank_sharded_raw_vectors_host_mapped_and_local_arrays(oracle):
    """C5's shape: raw vectors stay in host memory (page-locked, gathered over PCIe by the refine kernel) and every
    rank owns only its partitions' rows — handles cut out of the global arrays and handles built from local arrays."""
    from sharded_model import shard_local
    world, dim, m = 3, 64, 32
    rng = np.random.default_rng(11)
    s = train.synthetic_index(30000, dim, 40, m, seed=8, skew=0.9, empty_parts=2)
    s["raw"] = rng.normal(size=(30000, dim)).astype(np.float32)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=s["raw"],
                           metric="cosine")
    q = rng.normal(size=(21, dim)).astype(np.float32)
    kw = dict(k=10, nprobe_min=9, nprobe_max=9, refine_factor=10)
    exp = o.search(q, **kw)
    owner = lancedb_amd.shard_plan(s["part_offsets"], world)
    glob = [lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"],
                                   raw_vectors=s["raw"], metric="cosine", shard_count=world, shard_rank=r,
                                   raw_host_mapped=True) for r in range(world)]
    locs = [shard_local(s, owner, r) for r in range(world)]
    loc = [lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], locs[r]["codes"], locs[r]["row_ids"],
                                  raw_vectors=locs[r]["raw"], metric="cosine", shard_count=world, shard_rank=r,
                                  local_arrays=True, raw_host_mapped=True) for r in range(world)]
    for shards in (glob, loc):
        comms = Comm.loopback(world)
        got = run_ranks([lambda r=r: ShardedSearcher(shards[r], comms[r]).search(q, _abi.make_params(**kw))
                         for r in range(world)])
        for r in range(world):
            _same(got[r], exp, f"rank {r}")


def test_c5_shape_eight_ranks_each_holding_its_raw_shard_in_hbm(oracle):
    """BASELINE.json configs[4] at its stated size (100 M x 1536 f32 = 614 GB) fits no single GPU, and no host this box
    offers: the way it runs is eight ranks, each with the raw rows of ITS partitions resident in HBM (77 GB per rank at
    full size).  Here at reduced rows, with C5's dimension, PQ width, metric and refine factor: every rank opens a shard
    handle from local arrays only (codes, row ids and raw f32 rows of its own partitions), refines the merged candidates
    it owns, the second gather completes the list — and every rank's result is the UNSHARDED oracle's, bit for bit."""
    from sharded_model import shard_local
    world, dim, m, nlist, n = 8, 1536, 96, 64, 48_000
    rng = np.random.default_rng(55)
    s = train.synthetic_index(n, dim, nlist, m, seed=15, skew=0.6, empty_parts=1)
    s["raw"] = rng.normal(size=(n, dim)).astype(np.float32)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=s["raw"], metric="cosine")
    q = (s["centroids"][rng.integers(0, nlist, size=33)] + rng.normal(0, 0.4, size=(33, dim))).astype(np.float32)
    owner = lancedb_amd.shard_plan(s["part_offsets"], world)
    locs = [shard_local(s, owner, r) for r in range(world)]
    shards = [lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], locs[r]["codes"], locs[r]["row_ids"],
                                     raw_vectors=locs[r]["raw"], metric="cosine", shard_count=world, shard_rank=r, local_arrays=True)
              for r in range(world)]
    assert sum(sh.info()[0] for sh in shards) == n
    for kw in (dict(k=10, nprobe_min=16, nprobe_max=16, refine_factor=10), dict(k=10, nprobe_min=64, nprobe_max=64, refine_factor=10)):
        exp = o.search(q, **kw)
        comms = Comm.loopback(world)
        got = run_ranks([lambda r=r: ShardedSearcher(shards[r], comms[r]).search(q, _abi.make_params(**kw)) for r in range(world)])
        for r in range(world):
            _same(got[r], exp, f"rank {r} {kw}")


def test_flat_rows_sharded_over_three_loopback_ranks(oracle):
    world = 3
    rng = np.random.default_rng(9)
    v = rng.normal(size=(30000, 64)).astype(np.float32)
    rid = rng.permutation(30000).astype(np.uint64) + 7
    cuts = [0, 9000, 9001, 30000]  # ragged slices, one of a single row
    flats = [lancedb_amd.FlatIndex(v[cuts[r]:cuts[r + 1]], row_ids=rid[cuts[r]:cuts[r + 1]]) for r in range(world)]
    comms = Comm.loopback(world)
    q = rng.normal(size=(40, 64)).astype(np.float32)
    for kw in (dict(k=10), dict(k=300, metric=_abi.METRIC_COSINE), dict(k=5, metric=_abi.METRIC_DOT)):
        kw2 = dict(nprobe_min=1, nprobe_max=1, **kw)
        exp = oracle.flat_search(v, q, row_ids=rid, **kw)
        got = run_ranks([lambda r=r: ShardedFlatSearcher(flats[r], comms[r]).search(q, _abi.make_params(**kw2))
                         for r in range(world)])
        for r in range(world):
            _same(got[r], exp, f"rank {r} {kw}")


def test_a_rank_that_goes_away_fails_its_peers_instead_of_hanging(oracle):
    world = 2
    s = train.synthetic_index(8000, 32, 16, 8, seed=2)
    shards = _shards(s, None, world)
    comms = Comm.loopback(world)
    q = np.random.default_rng(0).normal(size=(4, 32)).astype(np.float32)
    err = []

    def rank0():
        try:
            ShardedSearcher(shards[0], comms[0]).search(q, _abi.make_params(k=5, nprobe_min=4, nprobe_max=4))
        except lancedb_amd.EngineError as e:
            err.append(str(e))
    t = threading.Thread(target=rank0)
    t.start()
    import time
    time.sleep(0.5)      # rank 0 now waits inside the gather for rank 1, which never calls
    comms[1].close()     # ... and goes away: the group is aborted
    t.join(timeout=30)
    assert not t.is_alive() and err and "loopback" in err[0]
    # the communicator of the surviving rank reports the failure on every later call
    with pytest.raises(lancedb_amd.EngineError, match="cannot be used again|aborted"):
        ShardedSearcher(shards[0], comms[0]).search(q, _abi.make_params(k=5, nprobe_min=4, nprobe_max=4))


def test_shard_handle_and_communicator_must_agree(oracle):
    s = train.synthetic_index(8000, 32, 16, 8, seed=2)
    shards = _shards(s, None, 2)
    comms = Comm.loopback(3)
    q = np.zeros((1, 32), np.float32)
    with pytest.raises(lancedb_amd.InvalidInput, match="shard 0 of 2"):
        ShardedSearcher(shards[0], comms[0]).search(q, _abi.make_params(k=5, nprobe_min=4, nprobe_max=4))


def test_shards_cut_by_a_probe_weighted_plan_return_the_unsharded_result(oracle):
    """mi355_index_desc.part_owner: ownership from mi355_shard_plan_weighted over observed probe counts (the plan
    that balances the rows scanned, not the rows held) — results are the unsharded ones, whoever owns what."""
    world, dim, m = 3, 128, 32
    rng = np.random.default_rng(21)
    s = train.synthetic_index(50000, dim, 64, m, seed=15, skew=0.9, empty_parts=2)
    raw = rng.normal(size=(50000, dim)).astype(np.float32)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw)
    q = (s["centroids"][rng.integers(0, 8, size=64)] + rng.normal(0, 0.3, size=(64, dim))).astype(np.float32)  # a skewed query load
    probes = np.stack([o.select_probes(o.coarse(x), 8) for x in q])
    hits = np.bincount(probes.ravel().astype(np.int64), minlength=64).astype(np.float32)
    owner = lancedb_amd.shard_plan(s["part_offsets"], world, weights=hits)
    assert (owner != lancedb_amd.shard_plan(s["part_offsets"], world)).any()
    shards = [lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw,
                                     shard_count=world, shard_rank=r, part_owner=owner) for r in range(world)]
    assert sum(sh.info()[0] for sh in shards) == 50000
    comms = Comm.loopback(world)
    for kw in (dict(k=10, nprobe_min=8, nprobe_max=8), dict(k=10, nprobe_min=8, nprobe_max=8, refine_factor=5)):
        exp = o.search(q, **kw)
        got = run_ranks([lambda r=r: ShardedSearcher(shards[r], comms[r]).search(q, _abi.make_params(**kw)) for r in range(world)])
        for r in range(world):
            _same(got[r], exp, f"rank {r} {kw}")
    st = comms[0].stats()
    assert st["imbalance"] < 1.35  # the unweighted plan on this load: one shard scans most of the rows
    with pytest.raises(lancedb_amd.InvalidInput, match="not a shard"):
        lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], shard_count=world,
                               shard_rank=0, part_owner=np.full(64, 7, np.uint32))
