"""dev tool: one index per (m, dsub) pair given on the command line ("128x16 192x16 ..."), searched with the engine and
the oracle; prints what differs.  python tests/tools/dbg_widths.py 128x16 [k=10] [nq=7]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import lancedb_amd
from oracle import oracle as orc, train

orc.build()
shapes = [a for a in sys.argv[1:] if "x" in a]
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
k, nq = int(opts.get("k", 10)), int(opts.get("nq", 7))
for sh in shapes:
    m, dsub = (int(x) for x in sh.split("x"))
    dim, nlist = m * dsub, 12
    lens = np.array([0, 1, 63, 64, 65, 1024, 1025, 3000, 0, 5000, 17, 2048], dtype=np.int64)
    s = train.synthetic_index(int(lens.sum()), dim, nlist, m, seed=m)
    s["part_offsets"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    rng = np.random.default_rng(m)
    q = (s["centroids"][rng.integers(0, nlist, size=nq)] + rng.normal(0, 0.5, size=(nq, dim))).astype(np.float32)
    g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    o = orc.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    for nprobe in (1, 12):
        got = g.search(q, k=k, nprobe_min=nprobe, nprobe_max=nprobe)
        ids, dist, cnt, st = o.search(q, k=k, nprobe_min=nprobe, nprobe_max=nprobe)
        ok = (got.counts == cnt).all() and (got.rowids == ids).all() and (got.distances == dist).all()
        print(sh, "nprobe", nprobe, "variant", g.stats()["scan_variant"], "OK" if ok else "DIFFERS", flush=True)
        if not ok:
            print(" counts", got.counts[:4], cnt[:4])
            print(" ids  ", got.rowids[0, :6], ids[0, :6])
            print(" dist ", got.distances[0, :6], dist[0, :6])
