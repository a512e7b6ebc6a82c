"""world_size-2 (and 3) CPU runs of the multi-GPU glue over the gloo backend.

What is under test is the rank-level logic of lancedb_amd/distributed.py and the
shard plan of the C ABI (pure host code, no GPU needed): every rank must derive
the same partition ownership, the gathered [world, B, k] layout must be what the
reducer expects, and the merged result must equal the unsharded search.  The
per-shard scan and the k-way merge, which are HIP kernels in the product, are
stood in for by the CPU oracle here (test infrastructure); the same identity is
checked on the GPU with real shard handles in test_gpu_parity.py.
"""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

import lancedb_amd  # noqa: E402
from lancedb_amd import _abi  # noqa: E402
from lancedb_amd.distributed import ShardedSearcher  # noqa: E402
from lancedb_amd.index import SearchResult  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shard_local(s, owner, rank):
    """The rows a shard handle keeps: partitions it does not own become empty."""
    po = s["part_offsets"].astype(np.int64)
    keep = [np.arange(po[p], po[p + 1]) for p in range(len(po) - 1) if owner[p] == rank]
    rows = np.concatenate(keep) if keep else np.zeros(0, np.int64)
    lens = np.array([(po[p + 1] - po[p]) if owner[p] == rank else 0 for p in range(len(po) - 1)])
    out = dict(s)
    out["part_offsets"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    out["codes"] = np.ascontiguousarray(s["codes"][rows])
    out["row_ids"] = np.ascontiguousarray(s["row_ids"][rows])
    return out


class _OracleShard:
    """Stands in for IvfPqIndex(shard_count, shard_rank) on a CPU-only box."""

    def __init__(self, s, world, rank):
        from oracle import oracle as orc
        owner = lancedb_amd.shard_plan(s["part_offsets"], world)  # C ABI, host code
        loc = _shard_local(s, owner, rank)
        self.ox = orc.OracleIndex(loc["centroids"], loc["codebook"], loc["part_offsets"], loc["codes"], loc["row_ids"])
        self.rows = int(loc["part_offsets"][-1])

    def search(self, queries, params, out=None):
        ids, d, c, st = self.ox.search(queries, params)
        assert st == 0
        return SearchResult(torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(d),
                            torch.from_numpy(c.astype(np.int32)))

    # two-phase search, restated with the oracle's stage-wise entry points
    nlist = property(lambda self: self.ox.nlist)

    def coarse_topn(self, queries, nprobe, lo, hi):
        nq = len(queries)
        ids = np.full((nq, nprobe), np.iinfo(np.int64).max, np.int64)
        ids[:] = -1  # UINT64_MAX as int64
        d = np.full((nq, nprobe), np.inf, np.float32)
        n_sel = min(nprobe, hi - lo)
        for i, q in enumerate(queries):
            co = self.ox.coarse(q)[lo:hi]
            order = np.lexsort((np.arange(lo, hi), co))[:n_sel]
            ids[i, :n_sel] = lo + order
            d[i, :n_sel] = co[order]
        return torch.from_numpy(ids), torch.from_numpy(d), torch.full((nq,), n_sel, dtype=torch.int32)

    def search_probes(self, queries, probes, params, out=None):
        k, nq = params.k, len(queries)
        ids = np.full((nq, k), -1, np.int64)
        d = np.full((nq, k), np.inf, np.float32)
        cnt = np.zeros(nq, np.int32)
        po = self.ox.part_offsets.astype(np.int64)
        for i, q in enumerate(queries):
            cd, ci = [], []
            for p in probes[i].numpy().astype(np.int64):
                if po[p + 1] > po[p]:
                    cd.append(self.ox.adc_partition(self.ox.build_lut(q, int(p)), int(p)))
                    ci.append(self.ox.row_ids[po[p]:po[p + 1]])
            if cd:
                cd, ci = np.concatenate(cd), np.concatenate(ci)
                order = np.lexsort((ci, cd))[:k]
                n = len(order)
                ids[i, :n], d[i, :n], cnt[i] = ci[order].astype(np.int64), cd[order], n
        return SearchResult(torch.from_numpy(ids), torch.from_numpy(d), torch.from_numpy(cnt))


def _oracle_merge(g_ids, g_dist, g_cnt, k, stream=0):
    from oracle import oracle as orc
    ids, d, c = orc.merge_topk(g_ids.numpy().astype(np.uint64), g_dist.numpy(), g_cnt.numpy().astype(np.uint32), k)
    return torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(d), torch.from_numpy(c.astype(np.int32))


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from oracle import train
        s = train.synthetic_index(20000, 32, 24, 8, seed=11, skew=1.0, empty_parts=3)
        q = np.random.default_rng(5).normal(size=(19, 32)).astype(np.float32)
        shard = _OracleShard(s, world, rank)
        searcher = ShardedSearcher(shard, merge=_oracle_merge)
        two_phase = ShardedSearcher(shard, merge=_oracle_merge, shard_coarse=True)
        total = torch.tensor([shard.rows])
        dist.all_reduce(total)
        assert int(total) == 20000  # the plan is a partition of the rows
        full = orc.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
        for k, nprobe in ((10, 6), (1, 24), (40, 3)):
            p = _abi.make_params(k=k, nprobe_min=nprobe, nprobe_max=nprobe)
            got = searcher.search(q, p)
            ids, d, c, _ = full.search(q, p)
            assert (got.rowids.numpy().astype(np.uint64) == ids).all()
            assert (got.distances.numpy() == d).all()
            assert (got.counts.numpy().astype(np.uint32) == c).all()
            if nprobe <= 24:  # two-phase: sharded coarse stage + one more all-gather, same result
                got2 = two_phase.search(q, p)
                assert (got2.rowids.numpy().astype(np.uint64) == ids).all()
                assert (got2.distances.numpy() == d).all()
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_search_over_gloo_matches_unsharded(world, oracle):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert sorted(ret.keys()) == list(range(world))


def test_shard_plan_is_balanced_and_matches_oracle(oracle):
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 5000, size=257)
    po = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    for shards in (1, 2, 4, 8):
        own = lancedb_amd.shard_plan(po, shards)
        assert (own == oracle.shard_plan(po, shards)).all()
        load = np.bincount(own, weights=lens, minlength=shards)
        assert load.max() - load.min() <= lens.max()
