"""Multi-GPU search: one process per GPU, IVF partitions sharded across ranks.

SURVEY.md §8e: a query's result is the top-k of the union of its probed
partitions, and partitions are scanned independently, so the partition list
shards with no data-path exchange until the very end: every rank scans the
probed partitions it owns (`IvfPqIndex(shard_count=world, shard_rank=rank)`,
the deterministic `mi355_shard_plan`), then ONE all-gather of the per-rank
`[B, k]` candidates (RCCL over xGMI when the tensors are on the GPU; 120 B per
query per rank at k = 10, latency-bound) and a k-way merge on every rank
(`mi355_merge_topk`).  The coarse quantiser is replicated.

`torch.distributed` is plumbing only; the reference has no collective at all
(no communication backend), this exchange is the engine's own.
"""
from .index import SearchResult, merge_topk


def coarse_slice(nlist, world, rank):
    """Centroid slice a rank scores in the two-phase search (contiguous, balanced)."""
    return (nlist * rank) // world, (nlist * (rank + 1)) // world


class ShardedSearcher:
    """Wraps one rank's shard handle.  `index.search(queries, params, out=...)`
    must return a SearchResult of tensors living where the process group's
    backend can reach them (CUDA tensors for nccl/RCCL)."""

    def __init__(self, index, group=None, merge=merge_topk, stream=0, shard_coarse=False):
        """shard_coarse: two-phase search (C4, nlist = 65536): every rank scores only its
        slice of the centroids; one extra all-gather of `nprobe` (distance, partition id)
        pairs per query per rank selects the global probe list before the scan."""
        import torch.distributed as dist
        self.index, self.group, self.merge, self.stream = index, group, merge, stream
        self.shard_coarse = shard_coarse
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._buf = None

    def _buffers(self, r):
        import torch
        B, k = r.distances.shape
        key = (B, k, r.distances.device)
        if self._buf is None or self._buf[0] != key:
            w = self.world
            # concatenated along dim 0 (the form every backend accepts), viewed [world, B, k] for the merge
            self._buf = (key, torch.empty((w * B, k), dtype=r.rowids.dtype, device=r.rowids.device),
                         torch.empty((w * B, k), dtype=r.distances.dtype, device=r.distances.device),
                         torch.empty((w * B,), dtype=r.counts.dtype, device=r.counts.device))
        return self._buf[1:]

    def search(self, queries, params, out=None):
        """-> SearchResult identical on every rank and to the unsharded search."""
        import torch.distributed as dist
        if self.shard_coarse and self.world > 1:
            r = self._search_two_phase(queries, params, out)
        else:
            r = self.index.search(queries, params, out=out)
        if self.world == 1:
            return r
        g_ids, g_dist, g_cnt = self._buffers(r)
        dist.all_gather_into_tensor(g_ids, r.rowids.contiguous(), group=self.group)
        dist.all_gather_into_tensor(g_dist, r.distances.contiguous(), group=self.group)
        dist.all_gather_into_tensor(g_cnt, r.counts.contiguous(), group=self.group)
        B, k = r.distances.shape
        w = self.world
        ids, d, c = self.merge(g_ids.view(w, B, k), g_dist.view(w, B, k), g_cnt.view(w, B), params.k,
                               stream=self.stream)
        return SearchResult(ids, d, c)

    def _search_two_phase(self, queries, params, out):
        import torch
        import torch.distributed as dist
        w, nprobe = self.world, params.nprobe_min
        lo, hi = coarse_slice(self.index.nlist, w, self.rank)
        ids, d, c = self.index.coarse_topn(queries, nprobe, lo, hi)
        B = d.shape[0]
        g_ids = torch.empty((w * B, nprobe), dtype=ids.dtype, device=ids.device)
        g_d = torch.empty((w * B, nprobe), dtype=d.dtype, device=d.device)
        g_c = torch.empty((w * B,), dtype=c.dtype, device=c.device)
        dist.all_gather_into_tensor(g_ids, ids.contiguous(), group=self.group)
        dist.all_gather_into_tensor(g_d, d.contiguous(), group=self.group)
        dist.all_gather_into_tensor(g_c, c.contiguous(), group=self.group)
        probes, _, _ = self.merge(g_ids.view(w, B, nprobe), g_d.view(w, B, nprobe), g_c.view(w, B), nprobe,
                                  stream=self.stream)
        return self.index.search_probes(queries, probes, params, out=out)
