import numpy as np, sys
sys.path.insert(0, '.')
import lancedb_amd
from lancedb_amd import _abi
from oracle import oracle as orc, train
s = train.synthetic_index(5000, 32, 16, 8, seed=5008, empty_parts=2)
g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
o = orc.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
q = np.random.default_rng(3).normal(size=(9, 32)).astype(np.float32)
print("lens", np.diff(s["part_offsets"].astype(np.int64)))
for nprobe in (1, 4):
    for k in (1, 10, 63, 64, 65, 70, 128, 129, 256):
        r = g.search(q, k=k, nprobe_min=nprobe, nprobe_max=nprobe)
        ids, dist, cnt, st = o.search(q, k=k, nprobe_min=nprobe, nprobe_max=nprobe)
        okc = (r.counts == cnt).all(); oki = (r.rowids == ids).all(); okd = (r.distances == dist).all()
        print(nprobe, k, okc, oki, okd, r.counts.tolist(), cnt.tolist())
        if not oki and k <= 70:
            b = int(np.argmax((r.rowids != ids).any(1)))
            print(" q", b, "got", r.rowids[b][:12], r.distances[b][:12]); print("  exp", ids[b][:12], dist[b][:12])
