import sys, numpy as np
sys.path.insert(0, ".")
import lancedb_amd
from oracle import oracle as orc, train
orc.build()
def run(m, dim, reps):
    rng = np.random.default_rng(m)
    nlist = 14
    lens = np.array([0, 1, 63, 64, 65, 1024, 1025, 3000, 0, 5000, 17, 2048, 8192, 700], dtype=np.int64)
    n = int(lens.sum())
    s = train.synthetic_index(n, dim, nlist, m, seed=m + 3)
    s["part_offsets"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], metric="l2")
    o = orc.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], metric="l2")
    q = (s["centroids"][rng.integers(0, nlist, size=37)] + rng.normal(0, 0.5, size=(37, dim))).astype(np.float32)
    bad = {"img": 0, "inline": 0}
    for nprobe, k in ((1, 10), (5, 1), (14, 10), (14, 64), (14, 100), (14, 128)):
        kw = dict(k=k, nprobe_min=nprobe, nprobe_max=nprobe)
        e = o.search(q, **kw)
        for rep in range(reps):
            for name, inline in (("img", False), ("inline", True)):
                g.configure(lut_inline=inline)
                a = g.search(q, **kw)
                ok = (a.counts == e[2]).all() and (a.rowids == e[0]).all() and (a.distances == e[1]).all()
                if not ok:
                    bad[name] += 1
                    if bad[name] <= 3:
                        qi = np.nonzero((a.rowids != e[0]).any(axis=1) | (a.counts != e[2]))[0]
                        print(f"  MISMATCH m{m} {name} nprobe {nprobe} k {k} rep {rep}: queries {qi[:8]} counts {a.counts[qi[:4]]} vs {e[2][qi[:4]]}", flush=True)
    print(f"m {m} dim {dim}: mismatches {bad} of {6 * reps} each", flush=True)
for m, dim in ((112, 1792), (48, 768), (192, 3072), (24, 384)):
    run(m, dim, int(sys.argv[1]) if len(sys.argv) > 1 else 25)
