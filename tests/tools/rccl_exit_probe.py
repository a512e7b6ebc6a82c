"""Dev: which teardown order of comm / index handles leaves a process that exits cleanly.
usage: python tests/tools/rccl_exit_probe.py            (runs every case in a fresh process)
       python tests/tools/rccl_exit_probe.py CASE       (one case, in this process)"""
import os
import subprocess
import sys

CASES = ["comm_only", "comm_then_index", "gc_at_exit", "flat", "refine", "coarse", "device_io", "mismatch", "big_k",
         "local_arrays", "flat_k300", "range"]


def run(case):
    sys.path.insert(0, ".")
    import numpy as np
    import lancedb_amd
    from lancedb_amd import _abi
    from lancedb_amd.distributed import Comm, ShardedFlatSearcher, ShardedSearcher, unique_id
    from oracle import train
    if case == "lib_only":
        lancedb_amd.device_count()
        return
    uid = unique_id()
    if case == "uid_only":
        return
    comm = Comm(uid, 0, 1)
    if case == "comm_only":
        comm.close()
        return
    rng = np.random.default_rng(1)
    if case == "flat":
        v = rng.normal(size=(20000, 64)).astype(np.float32)
        f = lancedb_amd.FlatIndex(v)
        ShardedFlatSearcher(f, comm).search(v[:8], _abi.make_params(k=5, nprobe_min=1, nprobe_max=1))
        comm.close()
        f.close()
        return
    if case == "flat_k300":
        v = rng.normal(size=(20000, 64)).astype(np.float32)
        f = lancedb_amd.FlatIndex(v, row_ids=rng.permutation(20000).astype(np.uint64) + 7)
        for kw in (dict(k=300, metric=_abi.METRIC_COSINE), dict(k=5, metric=_abi.METRIC_DOT)):
            ShardedFlatSearcher(f, comm).search(v[:40], _abi.make_params(nprobe_min=1, nprobe_max=1, **kw))
        return
    if case == "local_arrays":
        sys.path.insert(0, "tests")
        from sharded_model import shard_local
        s = train.synthetic_index(30000, 128, 40, 32, seed=5, skew=0.9, empty_parts=3)
        s["raw"] = rng.normal(size=(30000, 128)).astype(np.float32)
        owner = lancedb_amd.shard_plan(s["part_offsets"], 3)
        q = rng.normal(size=(17, 128)).astype(np.float32)
        for r in range(3):
            loc = shard_local(s, owner, r)
            a = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], loc["codes"], loc["row_ids"],
                                       raw_vectors=loc["raw"], shard_count=3, shard_rank=r, local_arrays=True)
            a.search(q, k=5, nprobe_min=9, nprobe_max=9, refine_factor=4)
        return
    s = train.synthetic_index(40000, 128, 48, 32, seed=13, skew=0.9, empty_parts=2)
    raw = rng.normal(size=(40000, 128)).astype(np.float32)
    ix = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw)
    q = rng.normal(size=(33, 128)).astype(np.float32)
    one = {"refine": dict(k=10, nprobe_min=8, nprobe_max=8, refine_factor=10), "big_k": dict(k=300, nprobe_min=6, nprobe_max=6),
           "range": dict(k=20, nprobe_min=2, nprobe_max=11, upper_bound=200.0)}
    if case in one:
        ShardedSearcher(ix, comm).search(q, _abi.make_params(**one[case]))
        return
    if case == "coarse":
        ShardedSearcher(ix, comm, shard_coarse=True).search(q, _abi.make_params(k=10, nprobe_min=16, nprobe_max=16))
        return
    if case == "device_io":
        DA = lancedb_amd.DeviceArray
        out = (DA((33, 10), np.int64), DA((33, 10), np.float32), DA((33,), np.int32))
        ShardedSearcher(ix, comm).search(DA.from_numpy(q), _abi.make_params(k=10, nprobe_min=16, nprobe_max=16), out=out)
        ix.sync()
        return
    if case == "mismatch":
        part = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], shard_count=2, shard_rank=0)
        try:
            ShardedSearcher(part, comm).search(q, _abi.make_params(k=5, nprobe_min=4, nprobe_max=4))
        except lancedb_amd.InvalidInput:
            pass
        return
    if case != "no_search":
        ShardedSearcher(ix, comm).search(q, _abi.make_params(k=5, nprobe_min=4, nprobe_max=4))
        comm.stats()
    if case in ("comm_then_index", "no_search"):
        comm.close()
        ix.close()
    elif case == "index_then_comm":
        ix.close()
        comm.close()
    # gc_at_exit: nothing closed explicitly


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
        print("case", sys.argv[1], "body done", flush=True)
    else:
        for env_name, env in (("torch-libs", {}),):
            for c in CASES:
                p = subprocess.run([sys.executable, __file__, c], env={**os.environ, **env}, capture_output=True, text=True, timeout=120)
                tail = [l for l in (p.stdout + p.stderr).strip().splitlines() if "amdgpu.ids" not in l and "Librccl" not in l][-2:]
                print(f"{env_name:12s} {c:16s} rc={p.returncode} {' | '.join(tail)[:200]}", flush=True)
