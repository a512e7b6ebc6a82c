"""Dev: exit behaviour of a process that initialises RCCL through the C ABI (mi355_comm_create).

Finding (round 2, gpurun calls r2g-r2i): a process that creates a communicator and only THEN
imports PyTorch (whose wheel ships the very librccl / libamdhip64 the engine pre-loaded) aborts at
exit with "double free or corruption" after all work completed correctly; importing torch first
(what bench.py does) or not at all is clean.  Every case runs in a fresh process.
usage: python tests/tools/rccl_exit_probe.py [CASE]"""
import os
import subprocess
import sys

CASES = ["comm_only", "torch_before_comm", "torch_after_comm", "search_then_exit"]


def run(case):
    sys.path.insert(0, ".")
    import numpy as np
    if case == "torch_before_comm":
        import torch  # noqa: F401
    import lancedb_amd
    from lancedb_amd import _abi
    from lancedb_amd.distributed import Comm, ShardedSearcher, unique_id
    from oracle import train
    comm = Comm(unique_id(), 0, 1)
    if case == "torch_after_comm":
        import torch  # noqa: F401
    if case == "search_then_exit":
        s = train.synthetic_index(20000, 32, 16, 8, seed=3)
        ix = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
        q = np.random.default_rng(1).normal(size=(9, 32)).astype(np.float32)
        ShardedSearcher(ix, comm).search(q, _abi.make_params(k=5, nprobe_min=4, nprobe_max=4))
        ix.close()
    comm.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
        print("case", sys.argv[1], "body done", flush=True)
    else:
        for c in CASES:
            p = subprocess.run([sys.executable, __file__, c], env=dict(os.environ), capture_output=True, text=True, timeout=200)
            lines = [l for l in (p.stdout + p.stderr).strip().splitlines() if "double free" in l or "body done" in l]
            print(f"{c:20s} rc={p.returncode} {' | '.join(lines)}", flush=True)
