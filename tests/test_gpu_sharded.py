"""The multi-GPU exchange through the real C path (csrc/ann_comm.hip: mi355_comm_*,
mi355_search_sharded, mi355_flat_search_sharded): RCCL behind the C ABI, no PyTorch.

A 1-GPU box can only form a world of one rank (RCCL refuses two ranks on one device), which
still runs every step of the path — slab packing, ncclAllGather on the handle's stream, merge
with owners, owner-side refine, second gather, maximum_nprobes second pass.  With >= 2 GPUs the
launcher test starts one torch-free process per GPU (id exchanged through a file) and checks the
result of every rank against the unsharded oracle.  The rank-level protocol itself is checked on
2 and 3 ranks on the CPU by tests/test_distributed_gloo.py (model of the same steps)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import lancedb_amd
from lancedb_amd import _abi
from lancedb_amd.distributed import Comm, ShardedFlatSearcher, ShardedSearcher, coarse_slice, unique_id
from oracle import train

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(got, exp):
    ids, dist, cnt, st = exp
    assert st == 0
    assert (np.asarray(got.counts) == cnt).all() and (np.asarray(got.rowids) == ids).all()
    assert (np.asarray(got.distances) == dist).all()


def test_coarse_slices_partition_the_centroids():
    for nlist, world in ((96, 5), (65536, 8), (3, 8), (4096, 1)):
        sl = [coarse_slice(nlist, world, r) for r in range(world)]
        assert sl[0][0] == 0 and sl[-1][1] == nlist and all(sl[i][1] == sl[i + 1][0] for i in range(world - 1))


@pytest.mark.parametrize("m,dim", [(8, 32), (32, 128)])
def test_world_of_one_runs_the_whole_exchange(oracle, m, dim):
    rng = np.random.default_rng(6)
    s = train.synthetic_index(40000, dim, 48, m, seed=13, skew=0.9, empty_parts=2)
    raw = rng.normal(size=(40000, dim)).astype(np.float32)
    ix = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw)
    comm = Comm(unique_id(), 0, 1)
    q = rng.normal(size=(33, dim)).astype(np.float32)
    cases = [dict(k=10, nprobe_min=16, nprobe_max=16), dict(k=10, nprobe_min=8, nprobe_max=8, refine_factor=10),
             dict(k=300, nprobe_min=6, nprobe_max=6), dict(k=7, nprobe_min=5, nprobe_max=5, refine_factor=50),
             dict(k=20, nprobe_min=2, nprobe_max=11, upper_bound=float(o.search(q, k=30, nprobe_min=11, nprobe_max=11)[1][0, 25]))]
    for coarse in (False, True):
        sh = ShardedSearcher(ix, comm, shard_coarse=coarse)
        for kw in cases:
            _same(sh.search(q, _abi.make_params(**kw)), o.search(q, **kw))
        st = comm.stats()
        assert st["world"] == 1 and st["n_gathers"] >= 1 and st["bytes_gathered"] > 0
        assert st["rows_scanned"][0] > 0 and abs(st["imbalance"] - 1.0) < 1e-6
    # device-resident I/O on the handle's own stream, no host synchronisation inside the call
    DA = lancedb_amd.DeviceArray
    dq = DA.from_numpy(q)
    out = (DA((33, 10), np.int64), DA((33, 10), np.float32), DA((33,), np.int32))
    r = ShardedSearcher(ix, comm).search(dq, _abi.make_params(k=10, nprobe_min=16, nprobe_max=16), out=out)
    ix.sync()
    ids, dist, cnt, _ = o.search(q, k=10, nprobe_min=16, nprobe_max=16)
    assert (r.rowids.numpy().view(np.uint64) == ids).all() and (r.distances.numpy() == dist).all()
    # a handle that is a shard of a bigger world does not match a world of one
    part = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], shard_count=2, shard_rank=0)
    with pytest.raises(lancedb_amd.InvalidInput, match="shard 0 of 2"):
        ShardedSearcher(part, comm).search(q, _abi.make_params(k=5, nprobe_min=4, nprobe_max=4))


def test_shard_handles_from_local_arrays_equal_those_from_global_arrays(oracle):
    """MI355_INDEX_LOCAL_ARRAYS: a rank hands over only the partitions it owns (bench.py generates
    exactly those); the handle must behave as the one cut out of the global arrays."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from sharded_model import shard_local
    rng = np.random.default_rng(3)
    for m, dim in ((8, 32), (32, 128)):
        s = train.synthetic_index(30000, dim, 40, m, seed=5, skew=0.9, empty_parts=3)
        s["raw"] = rng.normal(size=(30000, dim)).astype(np.float32)
        owner = lancedb_amd.shard_plan(s["part_offsets"], 3)
        q = rng.normal(size=(17, dim)).astype(np.float32)
        for r in range(3):
            loc = shard_local(s, owner, r)
            tr = train.to_part_transposed(loc["codes"], loc["part_offsets"])
            for codes, layout in ((loc["codes"], _abi.CODES_ROW_MAJOR), (tr, _abi.CODES_PART_TRANSPOSED)):
                a = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], codes, loc["row_ids"],
                                           raw_vectors=loc["raw"], codes_layout=layout, shard_count=3, shard_rank=r, local_arrays=True)
                b = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"],
                                           raw_vectors=s["raw"], shard_count=3, shard_rank=r)
                assert a.info() == b.info()
                for kw in (dict(k=10, nprobe_min=12, nprobe_max=12), dict(k=5, nprobe_min=9, nprobe_max=9, refine_factor=4)):
                    ra, rb = a.search(q, **kw), b.search(q, **kw)
                    assert (ra.rowids == rb.rowids).all() and (ra.distances == rb.distances).all() and (ra.counts == rb.counts).all()


def test_flat_rows_sharded_world_of_one(oracle):
    rng = np.random.default_rng(9)
    v = rng.normal(size=(20000, 64)).astype(np.float32)
    rid = rng.permutation(20000).astype(np.uint64) + 7
    f = lancedb_amd.FlatIndex(v, row_ids=rid)
    comm = Comm(unique_id(), 0, 1)
    q = rng.normal(size=(40, 64)).astype(np.float32)
    for kw in (dict(k=10), dict(k=300, metric=_abi.METRIC_COSINE), dict(k=5, metric=_abi.METRIC_DOT)):
        kw2 = dict(nprobe_min=1, nprobe_max=1, **kw)
        _same(ShardedFlatSearcher(f, comm).search(q, _abi.make_params(**kw2)), oracle.flat_search(v, q, row_ids=rid, **kw))


_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, {root!r})
import lancedb_amd
from lancedb_amd import _abi
from lancedb_amd.distributed import Comm, ShardedSearcher, ShardedFlatSearcher, exchange_id_via_file
from oracle import oracle as orc, train
rank, world, idfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
assert "torch" not in sys.modules
rng = np.random.default_rng(6)
s = train.synthetic_index(60000, 128, 64, 32, seed=13, skew=0.9, empty_parts=2)
raw = rng.normal(size=(60000, 128)).astype(np.float32)
q = rng.normal(size=(41, 128)).astype(np.float32)
ix = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw,
                            device=rank, shard_count=world, shard_rank=rank)
comm = Comm(exchange_id_via_file(idfile, rank), rank, world, device=rank)
o = orc.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw)
for coarse in (False, True):
    sh = ShardedSearcher(ix, comm, shard_coarse=coarse)
    for kw in (dict(k=10, nprobe_min=16, nprobe_max=16), dict(k=10, nprobe_min=8, nprobe_max=8, refine_factor=10),
               dict(k=300, nprobe_min=6, nprobe_max=6), dict(k=20, nprobe_min=2, nprobe_max=30, upper_bound=60.0)):
        got = sh.search(q, _abi.make_params(**kw))
        ids, dist, cnt, st = o.search(q, **kw)
        assert st == 0 and (got.counts == cnt).all() and (got.rowids == ids).all() and (got.distances == dist).all(), (kw, coarse)
st = comm.stats()
assert st["world"] == world and sum(st["rows_scanned"]) > 0 and st["imbalance"] >= 1.0
# flat: rows sharded across ranks
v = rng.normal(size=(30000, 64)).astype(np.float32)
lo, hi = 30000 * rank // world, 30000 * (rank + 1) // world
f = lancedb_amd.FlatIndex(v[lo:hi], row_ids=np.arange(lo, hi, dtype=np.uint64), device=rank)
fq = rng.normal(size=(20, 64)).astype(np.float32)
got = ShardedFlatSearcher(f, comm).search(fq, _abi.make_params(k=10, nprobe_min=1, nprobe_max=1))
ids, dist, cnt, _ = orc.flat_search(v, fq, k=10)
assert (got.rowids == ids).all() and (got.distances == dist).all()
print("rank", rank, "ok", json.dumps(st))
'''


def test_torch_free_launcher_one_process_per_gpu(oracle, tmp_path):
    world = min(lancedb_amd.device_count(), 4)
    if world < 2:
        pytest.skip("needs >= 2 GPUs: RCCL refuses two ranks on one device (the world-of-one test covers the path)")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    idfile = str(tmp_path / "comm_id")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), idfile], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}:\n{out[-3000:]}"
        assert f"rank {r} ok" in out
