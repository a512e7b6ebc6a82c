"""Dev: stress of the sliced-batch path (by-rows planner, LAT scan kernels, one-row-per-thread merge) against the CPU oracle:
random batch sizes 1..8, k in {1, 10, 16, 17, 50, 100, 128}, nprobes in {1, 5, 12, 40}, refine / ranges now and then; every result `==`.
usage: python tests/tools/lat_stress.py [searches] [m:dim ...]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import lancedb_amd  # noqa: E402
from oracle import oracle as orc, train  # noqa: E402

orc.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
SHAPES = [tuple(int(x) for x in a.split(":")) for a in sys.argv[2:]] or [(96, 768), (48, 768), (32, 128), (80, 320)]
for m, dim in SHAPES:
    rng = np.random.default_rng(m + dim)
    n, nlist = 600_000, 40
    s = train.synthetic_index(n, dim, nlist, m, seed=4, skew=1.2, empty_parts=2)
    raw = rng.normal(size=(n, dim)).astype(np.float32)
    ix = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw)
    o = orc.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw)
    ix.configure(graph=False, coalesce=False)
    bad = 0
    for it in range(N):
        nq = int(rng.integers(1, 9))
        k = int(rng.choice([1, 10, 16, 17, 50, 100, 128]))
        nprobe = int(rng.choice([1, 5, 12, 40]))
        q = (s["centroids"][rng.integers(0, nlist, size=nq)] + rng.normal(0, 0.5, size=(nq, dim))).astype(np.float32)
        kw = dict(k=k, nprobe_min=nprobe, nprobe_max=nprobe)
        if it % 7 == 3 and k <= 12:
            kw["refine_factor"] = 10
        e = o.search(q, **kw)
        a = ix.search(q, **kw)
        ok = (a.counts == e[2]).all() and (a.rowids == e[0]).all() and (a.distances == e[1]).all()
        if not ok:
            bad += 1
            if bad <= 5:
                print(f"  MISMATCH m {m} dim {dim}: search {it} nq {nq} {kw}", flush=True)
    print(f"m {m} dim {dim}: {bad} mismatches in {N} searches", flush=True)
