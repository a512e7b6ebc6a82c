"""Dev: for the intermittent wrong result of the table-image path with SK_IMG_PREFETCH=1 (NOTES 11.11): repeats one 520-query search until it
differs from the oracle and asks the oracle for which OTHER query of the batch every wrong (row, distance) would have been right.
usage: MI355_ANN_LIB=lancedb_amd/variants/lib_pf1.so python tests/tools/image_race_diag.py"""
import sys, numpy as np
sys.path.insert(0, ".")
import lancedb_amd
from oracle import oracle as orc, train
orc.build()
m, dim = 192, 3072
rng = np.random.default_rng(m)
nlist = 14
lens = np.array([0, 1, 63, 64, 65, 1024, 1025, 3000, 0, 5000, 17, 2048, 8192, 700], dtype=np.int64)
n = int(lens.sum())
s = train.synthetic_index(n, dim, nlist, m, seed=m + 3)
s["part_offsets"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
po = s["part_offsets"].astype(np.int64)
g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], metric="l2")
o = orc.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], metric="l2")
NQ = 520
q = (s["centroids"][rng.integers(0, nlist, size=NQ)] + rng.normal(0, 0.5, size=(NQ, dim))).astype(np.float32)
kw = dict(k=10, nprobe_min=14, nprobe_max=14)
e = o.search(q, **kw)
rid = np.asarray(s["row_ids"]).astype(np.uint64)
pos_of = {int(r): i for i, r in enumerate(rid)}
found = 0
for rep in range(200):
    a = g.search(q, **kw)
    bad = np.nonzero((a.rowids != e[0]).any(axis=1) | (a.distances != e[1]).any(axis=1))[0]
    if not len(bad):
        continue
    found += 1
    print(f"rep {rep}: wrong queries {bad[:12]}", flush=True)
    for b in bad[:3]:
        exp = {int(i): float(d) for i, d in zip(e[0][b], e[1][b])}
        wrong = [(int(i), float(d)) for i, d in zip(a.rowids[b], a.distances[b]) if exp.get(int(i)) != float(d)]
        print(f"  query {b}: wrong (id, d): {wrong[:4]}", flush=True)
        for rid_w, d_w in wrong[:2]:
            p_row = pos_of.get(rid_w, -1)
            part = int(np.searchsorted(po, p_row, side="right") - 1) if p_row >= 0 else -1
            # the row's TRUE distance for this query, and for which OTHER query the returned value is the true one
            allow = np.array([rid_w], dtype=np.uint64)
            all_d = o.search(q, k=1, nprobe_min=14, nprobe_max=14, allow_rowids=allow)
            dd = all_d[1][:, 0]
            match = np.nonzero(dd == np.float32(d_w))[0]
            print(f"    row id {rid_w} (position {p_row}, partition {part}, length {lens[part] if part >= 0 else -1}): true d for query {b} = {dd[b]}, returned {d_w}; the returned value is the true distance of queries {match[:6]}", flush=True)
    if found >= 3:
        break
print("searches with mismatches:", found)
