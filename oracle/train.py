"""Small numpy IVF-PQ trainer: builds REAL indexes for parity and recall tests.

TEST INFRASTRUCTURE ONLY (same rule as oracle.py).  It follows the parameter
meanings of the reference's builder — num_partitions, num_sub_vectors, 8 bits,
max_iterations, sample_rate (rust/lancedb/src/index/vector.rs:266-319,
table/create_index.rs:283-303) — with plain Lloyd k-means; lance's own trainer
([EXT]) is not reproduced, only the layout of what the scan reads:
centroids [nlist, dim], codebook [m, 256, dim/m] trained on residuals for
L2/cosine, codes grouped by partition, row ids in index order.
"""
import numpy as np


def _sq_dists(x, c):
    return (x * x).sum(1)[:, None] - 2.0 * (x @ c.T) + (c * c).sum(1)[None, :]


def kmeans(x, k, iters, rng):
    n = x.shape[0]
    if n >= k:
        c = x[rng.choice(n, size=k, replace=False)].copy()
    else:
        c = np.concatenate([x, x[rng.integers(0, n, size=k - n)]
                            + rng.normal(0, 1e-3, size=(k - n, x.shape[1])).astype(x.dtype)])
    for _ in range(iters):
        a = _sq_dists(x, c).argmin(1)
        for j in range(k):
            sel = a == j
            if sel.any():
                c[j] = x[sel].mean(0)
            else:  # re-seed an empty cluster
                c[j] = x[rng.integers(0, n)]
    return c.astype(np.float32)


def train_ivfpq(vectors, nlist, m, metric="l2", iters=8, seed=0, row_ids=None):
    """-> dict(centroids, codebook, part_offsets, codes (row-major, index order),
    row_ids (index order), raw (index order), assign)."""
    rng = np.random.default_rng(seed)
    x = np.ascontiguousarray(vectors, dtype=np.float32)
    n, dim = x.shape
    assert dim % m == 0
    dsub = dim // m
    xt = x
    if metric == "cosine":
        nrm = np.linalg.norm(x, axis=1, keepdims=True)
        nrm[nrm == 0] = 1.0
        xt = (x / nrm).astype(np.float32)
    centroids = kmeans(xt, nlist, iters, rng)
    assign = _sq_dists(xt, centroids).argmin(1) if metric != "dot" else (-(xt @ centroids.T)).argmin(1)
    resid = xt - centroids[assign] if metric != "dot" else xt
    codebook = np.empty((m, 256, dsub), dtype=np.float32)
    codes = np.empty((n, m), dtype=np.uint8)
    for j in range(m):
        sub = np.ascontiguousarray(resid[:, j * dsub:(j + 1) * dsub])
        cb = kmeans(sub, 256, max(2, iters // 2), rng)
        codebook[j] = cb
        if metric == "dot":
            codes[:, j] = (-(sub @ cb.T)).argmin(1)
        else:
            codes[:, j] = _sq_dists(sub, cb).argmin(1)
    order = np.argsort(assign, kind="stable")
    counts = np.bincount(assign, minlength=nlist)
    part_offsets = np.zeros(nlist + 1, dtype=np.uint64)
    part_offsets[1:] = np.cumsum(counts)
    rid = np.arange(n, dtype=np.uint64) if row_ids is None else np.asarray(row_ids, dtype=np.uint64)
    return dict(centroids=centroids, codebook=codebook, part_offsets=part_offsets,
                codes=np.ascontiguousarray(codes[order]), row_ids=np.ascontiguousarray(rid[order]),
                raw=np.ascontiguousarray(x[order]), assign=assign[order])


def synthetic_index(n, dim, nlist, m, seed=0, skew=0.5, empty_parts=0, nbits=8):
    """Random (untrained) index with the shape of the throughput datasets of
    SURVEY.md §8d: N(0,1) centroids, N(0,0.25) codebook, uniform u8 codes,
    log-normally skewed partition lengths, permuted row ids."""
    rng = np.random.default_rng(seed)
    dsub = dim // m
    centroids = rng.normal(0, 1, size=(nlist, dim)).astype(np.float32)
    codebook = rng.normal(0, 0.5, size=(m, 1 << nbits, dsub)).astype(np.float32)
    w = np.exp(rng.normal(0, skew, size=nlist))
    if empty_parts:
        w[rng.choice(nlist, size=empty_parts, replace=False)] = 0
    lens = rng.multinomial(n, w / w.sum())
    part_offsets = np.zeros(nlist + 1, dtype=np.uint64)
    part_offsets[1:] = np.cumsum(lens)
    codes = rng.integers(0, 256, size=(n, m * nbits // 8), dtype=np.uint8)  # 4-bit: two random nibbles per byte
    row_ids = rng.permutation(n).astype(np.uint64)
    return dict(centroids=centroids, codebook=codebook, part_offsets=part_offsets,
                codes=codes, row_ids=row_ids)


def to_part_transposed(codes, part_offsets):
    """row-major [n, m] (index order) -> lance's per-partition [m, len_p] blocks."""
    out = np.empty(codes.size, dtype=np.uint8)
    m = codes.shape[1]
    for p in range(len(part_offsets) - 1):
        o, e = int(part_offsets[p]), int(part_offsets[p + 1])
        out[o * m:e * m] = codes[o:e].T.reshape(-1)
    return out
