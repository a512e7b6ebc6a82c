import numpy as np, sys, os
sys.path.insert(0, '.')
import lancedb_amd
from oracle import oracle as orc, train
m = int(os.environ.get("M", 96)); nq = int(os.environ.get("NQ", 64)); n = int(os.environ.get("N", 200000))
s = train.synthetic_index(n, m * 8, 8, m, seed=5, skew=0.8)
g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
o = orc.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
q = (s["centroids"][np.random.default_rng(1).integers(0, 8, nq)] + np.random.default_rng(2).normal(0, 0.5, size=(nq, m * 8))).astype(np.float32)
r = g.search(q, k=10, nprobe_min=8, nprobe_max=8)
ids, dist, cnt, st = o.search(q, k=10, nprobe_min=8, nprobe_max=8)
print("M", m, "nq", nq, "ids", (r.rowids == ids).all(), "dist", (r.distances == dist).all(), flush=True)
