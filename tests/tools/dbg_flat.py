import numpy as np, sys, os
sys.path.insert(0, '.')
import lancedb_amd
from lancedb_amd import _abi
from oracle import oracle as orc
rng = np.random.default_rng(8)
metric = os.environ.get("METRIC", "dot")
v = rng.normal(size=(70000, 48)).astype(np.float32)
q = rng.normal(size=(5, 48)).astype(np.float32)
f = lancedb_amd.FlatIndex(v)
mt = _abi.METRIC_NAMES[metric]
for k in (1, 10, 100):
    r = f.search(q, k=k, metric=mt)
    ids, dist, cnt, st = orc.flat_search(v, q, k=k, metric=mt)
    print(metric, k, (r.rowids == ids).all(), (r.distances == dist).all(), f.info(), flush=True)
