"""Host side of `Table.create_index(Index::IvfPq(..))` for this engine: the
reference's builder parameters (rust/lancedb/src/index/vector.rs:61-119,
:306-319; mapping to IvfBuildParams / PQBuildParams in
rust/lancedb/src/table/create_index.rs:68-102, :283-303) driving the GPU
trainer and encoder of the C ABI (mi355_kmeans_train, mi355_ivf_residuals,
mi355_ivfpq_encode).  The sampling and the initial centroids are host logic
(numpy, seeded); every O(rows) step runs on the device.  No CPU fallback.
"""
import math

import numpy as np

from .index import IvfPqIndex, ivf_residuals, ivfpq_encode, kmeans_train


def suggested_num_partitions(n_rows):
    """sqrt(rows) (index/vector.rs:64-66 'By default the number of partitions is
    the square root of the number of rows')."""
    return max(1, int(math.sqrt(n_rows)))


def suggested_num_sub_vectors(dim):
    """index/vector.rs:306-319."""
    if dim % 16 == 0:
        return dim // 16
    if dim % 8 == 0:
        return dim // 8
    return 1


class IvfPqBuilder:
    """Mirror of `IvfPqIndexBuilder` (index/vector.rs:142-200): distance_type,
    num_partitions, sample_rate, max_iterations, num_sub_vectors, num_bits."""

    def __init__(self, distance_type="l2", num_partitions=None, num_sub_vectors=None, num_bits=8, sample_rate=256,
                 max_iterations=50, seed=0):
        if num_bits != 8:
            raise NotImplementedError("only 8-bit PQ codes")  # 4-bit: NotSupported throughout the engine
        self.distance_type, self.num_partitions, self.num_sub_vectors = distance_type, num_partitions, num_sub_vectors
        self.sample_rate, self.max_iterations, self.seed = sample_rate, max_iterations, seed

    def _sample(self, x, count, rng):
        if count >= x.shape[0]:
            return x
        return np.ascontiguousarray(x[np.sort(rng.choice(x.shape[0], size=count, replace=False))])

    def train(self, vectors):
        """-> (centroids [nlist, dim], codebook [m, 256, dim/m]) from a sample of
        `sample_rate * num_partitions` rows (IVF) / `sample_rate * 256` rows (PQ)."""
        x = np.ascontiguousarray(vectors, dtype=np.float32)
        n, dim = x.shape
        nlist = self.num_partitions or suggested_num_partitions(n)
        m = self.num_sub_vectors or suggested_num_sub_vectors(dim)
        if dim % m:
            raise ValueError(f"num_sub_vectors {m} does not divide the dimension {dim}")
        if n < max(nlist, 256):
            raise ValueError(f"not enough rows ({n}) to train {nlist} partitions / 256 PQ centroids")
        rng = np.random.default_rng(self.seed)
        metric = self.distance_type
        ivf_sample = self._sample(x, self.sample_rate * nlist, rng)
        init = ivf_sample[np.sort(rng.choice(ivf_sample.shape[0], size=nlist, replace=False))]
        if metric == "cosine":  # the trainer normalises the rows; seed it with normalised rows too
            init = init / np.maximum(np.linalg.norm(init, axis=1, keepdims=True), np.float32(1e-30))
        centroids, _ = kmeans_train(ivf_sample, init, metric=metric, iters=self.max_iterations)
        pq_sample = self._sample(x, self.sample_rate * 256, rng)
        resid, _ = ivf_residuals(pq_sample, centroids, metric=metric)
        dsub = dim // m
        codebook = np.empty((m, 256, dsub), dtype=np.float32)
        pick = np.sort(rng.choice(resid.shape[0], size=256, replace=False))
        # residuals are already normalised / centred: the sub-quantisers are plain L2 (dot: dot) k-means
        sub_metric = "dot" if metric == "dot" else "l2"
        for j in range(m):
            cols = (j * dsub, (j + 1) * dsub)
            codebook[j], _ = kmeans_train(resid, resid[pick, cols[0]:cols[1]], metric=sub_metric,
                                          iters=self.max_iterations, cols=cols)
        return centroids, codebook

    def build(self, vectors, row_ids=None, keep_vectors=True):
        """Train, encode every row and open the device index."""
        x = np.ascontiguousarray(vectors, dtype=np.float32)
        centroids, codebook = self.train(x)
        po, codes, order = ivfpq_encode(x, centroids, codebook, metric=self.distance_type)
        order = order.astype(np.int64)
        ids = order.astype(np.uint64) if row_ids is None else np.asarray(row_ids, dtype=np.uint64)[order]
        return IvfPqIndex(centroids, codebook, po, codes, ids, raw_vectors=x[order] if keep_vectors else None,
                          metric=self.distance_type)
