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

from .index import IvfPqIndex, ivf_residuals, ivfpq_encode, kmeans_train, pq_train


DEFAULT_PARTITION_SIZE = 8192  # rows per partition of IvfBuildParams::default() [EXT], pinned by
# table/create_index.rs:733-795 (test_ivf_pq_uses_default_partition_size_for_num_partitions: 2 * 8192 rows -> 2)


def num_partitions_for(n_rows, num_partitions=None, target_partition_size=None):
    """build_ivf_params (table/create_index.rs:66-84): an explicit num_partitions wins, then
    target_partition_size, then the default partition size: rows / 8192 (at least 1)."""
    if num_partitions:
        return int(num_partitions)
    size = int(target_partition_size) if target_partition_size else DEFAULT_PARTITION_SIZE
    return max(1, int(n_rows) // size)


def suggested_num_partitions(n_rows):
    """The engine-side default when nothing is given: rows / 8192 (see num_partitions_for)."""
    return num_partitions_for(n_rows)


def suggested_num_sub_vectors(dim):
    """index/vector.rs:306-319."""
    if dim % 16 == 0:
        return dim // 16
    if dim % 8 == 0:
        return dim // 8
    return 1


def get_num_sub_vectors(provided, dim, num_bits=None):
    """table/create_index.rs:86-102: the suggestion, made even when 4 bits are used."""
    if provided:
        return int(provided)
    m = suggested_num_sub_vectors(dim)
    if num_bits == 4 and m % 2:
        m += 1
    return m


class IvfPqBuilder:
    """Mirror of `IvfPqIndexBuilder` (index/vector.rs:61-119, :142-200, :266-304): distance_type,
    num_partitions, target_partition_size, sample_rate, max_iterations, num_sub_vectors,
    num_bits (8, or 4 with an even num_sub_vectors: table/create_index.rs:96-101)."""

    def __init__(self, distance_type="l2", num_partitions=None, num_sub_vectors=None, num_bits=8, sample_rate=256,
                 max_iterations=50, seed=0, target_partition_size=None):
        if num_bits not in (4, 8):
            raise ValueError(f"num_bits must be 4 or 8, got {num_bits}")
        if num_bits == 4 and num_sub_vectors and num_sub_vectors % 2:
            raise ValueError(f"num_sub_vectors must be even when num_bits is 4, got {num_sub_vectors}")
        self.distance_type, self.num_partitions, self.num_sub_vectors = distance_type, num_partitions, num_sub_vectors
        self.target_partition_size, self.num_bits = target_partition_size, num_bits
        self.sample_rate, self.max_iterations, self.seed = sample_rate, max_iterations, seed

    def _sample(self, x, count, rng):
        if count >= x.shape[0]:
            return x
        return np.ascontiguousarray(x[np.sort(rng.choice(x.shape[0], size=count, replace=False))])

    def train(self, vectors):
        """-> (centroids [nlist, dim], codebook [m, 2^num_bits, dim/m]) from a sample of
        `sample_rate * num_partitions` rows (IVF) / `sample_rate * 2^num_bits` rows (PQ)."""
        x = np.ascontiguousarray(vectors, dtype=np.float32)
        n, dim = x.shape
        nlist = num_partitions_for(n, self.num_partitions, self.target_partition_size)
        m = get_num_sub_vectors(self.num_sub_vectors, dim, self.num_bits)
        ks = 1 << self.num_bits
        if dim % m:
            raise ValueError(f"num_sub_vectors {m} does not divide the dimension {dim}")
        if n < max(nlist, ks):
            raise ValueError(f"not enough rows ({n}) to train {nlist} partitions / {ks} PQ centroids")
        rng = np.random.default_rng(self.seed)
        metric = self.distance_type
        ivf_sample = self._sample(x, self.sample_rate * nlist, rng)
        init = ivf_sample[np.sort(rng.choice(ivf_sample.shape[0], size=nlist, replace=False))]
        if metric == "cosine":  # the trainer normalises the rows; seed it with normalised rows too
            init = init / np.maximum(np.linalg.norm(init, axis=1, keepdims=True), np.float32(1e-30))
        centroids, _ = kmeans_train(ivf_sample, init, metric=metric, iters=self.max_iterations)
        pq_sample = self._sample(x, self.sample_rate * ks, rng)
        resid, _ = ivf_residuals(pq_sample, centroids, metric=metric)
        dsub = dim // m
        pick = np.sort(rng.choice(resid.shape[0], size=ks, replace=False))
        # the seeds of sub-quantiser j: its column range of `ks` sampled residuals.  Residuals are already
        # normalised / centred: the sub-quantisers are plain L2 (dot: dot) k-means, all m in one call
        init = np.ascontiguousarray(resid[pick].reshape(ks, m, dsub).transpose(1, 0, 2))
        codebook = pq_train(resid, init, metric=metric, iters=self.max_iterations, nbits=self.num_bits)
        return centroids, codebook

    def build(self, vectors, row_ids=None, keep_vectors=True):
        """Train, encode every row and open the device index."""
        x = np.ascontiguousarray(vectors, dtype=np.float32)
        centroids, codebook = self.train(x)
        po, codes, order = ivfpq_encode(x, centroids, codebook, metric=self.distance_type, nbits=self.num_bits)
        order = order.astype(np.int64)
        ids = order.astype(np.uint64) if row_ids is None else np.asarray(row_ids, dtype=np.uint64)[order]
        return IvfPqIndex(centroids, codebook, po, codes, ids, raw_vectors=x[order] if keep_vectors else None,
                          metric=self.distance_type, nbits=self.num_bits)
