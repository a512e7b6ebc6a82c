"""Dev: which teardown order of comm / index handles leaves a process that exits cleanly.
usage: python tests/tools/rccl_exit_probe.py            (runs every case in a fresh process)
       python tests/tools/rccl_exit_probe.py CASE       (one case, in this process)"""
import os
import subprocess
import sys

CASES = ["lib_only", "uid_only", "comm_only", "comm_then_index", "index_then_comm", "gc_at_exit", "flat", "no_search"]


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
    s = train.synthetic_index(20000, 32, 16, 8, seed=3)
    ix = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    q = rng.normal(size=(9, 32)).astype(np.float32)
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
        for env_name, env in (("torch-libs", {}), ("system-libs", {"MI355_HIP_RUNTIME": "system"})):
            for c in CASES:
                p = subprocess.run([sys.executable, __file__, c], env={**os.environ, **env}, capture_output=True, text=True, timeout=120)
                tail = (p.stdout + p.stderr).strip().splitlines()[-2:]
                print(f"{env_name:12s} {c:16s} rc={p.returncode} {' | '.join(tail)[:200]}", flush=True)
