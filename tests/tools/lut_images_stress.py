import sys, numpy as np
sys.path.insert(0, ".")
import lancedb_amd
from oracle import oracle as orc, train
orc.build()
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 37  # queries per search (520: the batch of tests/test_gpu_lut_images.py)
import ctypes as C
from lancedb_amd import _lib
L = _lib.lib()
HAS_DEV = hasattr(L, "mi355_dev_counters")  # (a -DSK_IMG_VERIFY2 build)
CNT = (C.c_uint32 * 8)()
def run(m, dim, reps):
    rng = np.random.default_rng(m)
    nlist = 14
    lens = np.array([0, 1, 63, 64, 65, 1024, 1025, 3000, 0, 5000, 17, 2048, 8192, 700], dtype=np.int64)
    n = int(lens.sum())
    s = train.synthetic_index(n, dim, nlist, m, seed=m + 3)
    s["part_offsets"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], metric="l2")
    o = orc.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], metric="l2")
    q = (s["centroids"][rng.integers(0, nlist, size=NQ)] + rng.normal(0, 0.5, size=(NQ, dim))).astype(np.float32)
    bad = {"img": 0, "inline": 0}
    dev_sum = {6: 0, 7: 0}
    for nprobe, k in ((1, 10), (5, 1), (14, 10), (14, 64), (14, 100), (14, 128)):
        kw = dict(k=k, nprobe_min=nprobe, nprobe_max=nprobe)
        e = o.search(q, **kw)
        for rep in range(reps):
            for name, inline in (("img", False), ("inline", True)):
                g.configure(lut_inline=inline)
                a = g.search(q, **kw)
                if HAS_DEV and not inline:
                    L.mi355_dev_counters(g._h, CNT, C.c_int32(1))
                    for x in (6, 7):
                        dev_sum[x] += CNT[x]
                    if CNT[6] or CNT[7]:
                        print(f"    LDS table != image: {CNT[6]} quads after the store, {CNT[7]} after the scan (nprobe {nprobe} k {k} rep {rep})", flush=True)
                ok = (a.counts == e[2]).all() and (a.rowids == e[0]).all() and (a.distances == e[1]).all()
                if not ok:
                    bad[name] += 1
                    if bad[name] <= 3:
                        qi = np.nonzero((a.rowids != e[0]).any(axis=1) | (a.counts != e[2]) | (a.distances != e[1]).any(axis=1))[0]
                        print(f"  MISMATCH m{m} {name} nprobe {nprobe} k {k} rep {rep}: {len(qi)} queries {qi[:8]} counts {a.counts[qi[:4]]} vs {e[2][qi[:4]]}", flush=True)
                        b0 = qi[0]
                        print(f"    query {b0}: got ids {a.rowids[b0][:6]} dist {a.distances[b0][:6]}", flush=True)
                        print(f"    query {b0}: exp ids {e[0][b0][:6]} dist {e[1][b0][:6]}", flush=True)
    print(f"m {m} dim {dim}: mismatches {bad} of {6 * reps} each", flush=True)
    if HAS_DEV:
        print(f"   LDS table quads that differed from the image: {dev_sum[6]} after the store, {dev_sum[7]} after the scan", flush=True)
SHAPES = [tuple(int(x) for x in a.split(':')) for a in sys.argv[3:]] or [(192, 3072), (96, 1536), (48, 768)]
for m, dim in SHAPES:
    run(m, dim, int(sys.argv[1]) if len(sys.argv) > 1 else 25)
