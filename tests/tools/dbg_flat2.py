import numpy as np, sys, os
sys.path.insert(0, '.')
import lancedb_amd
from lancedb_amd import _abi
from oracle import oracle as orc
rng = np.random.default_rng(17)
n, dim = 20000, 80
v = rng.normal(size=(n, dim)).astype(np.float32)
v[100:4100] = v[100]
v[5000:5200] = v[5000] + rng.normal(0, 1e-4, size=(200, dim)).astype(np.float32)
v[6000] = 0.0
v[6001, 3] = np.nan
v[6002] = 3e38
q = np.concatenate([v[[100, 5000, 6000, 6002]], rng.normal(size=(127, dim)).astype(np.float32)])
f = lancedb_amd.FlatIndex(v)
for metric in ("l2", "cosine", "dot"):
    mt = _abi.METRIC_NAMES[metric]
    for k in (1, 10, 200):
        r = f.search(q, k=k, metric=mt)
        ids, dist, cnt, st = orc.flat_search(v, q, k=k, metric=mt)
        badc = np.nonzero(r.counts != cnt)[0]
        badi = np.nonzero((r.rowids != ids).any(1))[0]
        print(metric, k, "count mismatches", badc[:8], "id mismatches", badi[:8], f.info(), flush=True)
        for b in list(badc[:2]) + list(badi[:2]):
            print("   q", b, "got cnt", r.counts[b], "exp", cnt[b], "got", r.rowids[b][:5], r.distances[b][:5], "exp", ids[b][:5], dist[b][:5])
