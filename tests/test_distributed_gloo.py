"""world_size-2 (and 3) CPU runs of the sharded-search PROTOCOL over the gloo backend.

The product's exchange is C++ + RCCL behind the C ABI (csrc/ann_comm.hip:
mi355_search_sharded) and needs GPUs; its rank-level protocol — shard plan, centroid
slices, the packed slab all-gather, merge with owners, owner-side refine, second gather,
collective maximum_nprobes decision — is restated in tests/sharded_model.py with the CPU
oracle standing in for the HIP kernels, and checked here against the UNSHARDED oracle
search on 2 and 3 ranks.  The shard plan and the slices come from the product's own host
code (mi355_shard_plan / mi355_coarse_slice, no GPU needed).  The same identities are
checked on the GPU through the real C path in tests/test_gpu_sharded.py.
"""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import lancedb_amd  # noqa: E402
from lancedb_amd import _abi  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sharded_model as sm
        from oracle import oracle as orc
        from oracle import train
        s = train.synthetic_index(20000, 32, 24, 8, seed=11, skew=1.0, empty_parts=3)
        rng = np.random.default_rng(5)
        s["raw"] = rng.normal(size=(20000, 32)).astype(np.float32)  # raw vectors in index order (refine)
        q = rng.normal(size=(19, 32)).astype(np.float32)
        shard = sm.OracleShard(s, world, rank)
        total = torch.tensor([shard.rows])
        dist.all_reduce(total)
        assert int(total) == 20000  # the plan is a partition of the rows
        full = orc.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=s["raw"])
        cases = [dict(k=10, nprobe_min=6, nprobe_max=6), dict(k=1, nprobe_min=24, nprobe_max=24),
                 dict(k=40, nprobe_min=3, nprobe_max=3),
                 dict(k=10, nprobe_min=5, nprobe_max=5, refine_factor=4),           # owner-side refine
                 dict(k=300, nprobe_min=4, nprobe_max=4),                          # k beyond one selection pass
                 dict(k=7, nprobe_min=4, nprobe_max=4, refine_factor=50),          # kk = 350
                 dict(k=10, nprobe_min=2, nprobe_max=9, upper_bound=16.0),         # short queries -> np_max together
                 dict(k=10, nprobe_min=2, nprobe_max=None, refine_factor=3, upper_bound=40.0)]
        for kw in cases:
            p = _abi.make_params(**kw)
            ids, d, c, st = full.search(q, p)
            assert st == 0
            for coarse in (False, True):
                g_ids, g_d, g_c = sm.sharded_search(shard, q, p, world, rank, shard_coarse=coarse)
                assert (g_c == c).all(), (kw, coarse)
                assert (g_ids == ids).all(), (kw, coarse)
                assert (g_d == d).all(), (kw, coarse)
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
