"""Rank-level MODEL of csrc/ann_comm.hip's sharded search, runnable on a CPU-only box.

Test infrastructure.  The product's exchange is C++ + RCCL behind the C ABI and needs a
GPU; what can be checked here is the PROTOCOL it implements, step for step, with the CPU
oracle standing in for the per-shard HIP kernels and torch.distributed (gloo) for RCCL:

  slab = [B x kk 16-byte records (distance f32, local position u32, rowid u64)] [B x u32
  counts, padded to 16 B] [16-byte trailer]          -> ONE all-gather of bytes per exchange
  1. (MI355_SHARD_COARSE) per-rank coarse lists over its centroid slice -> gather -> merge by
     (distance, partition id) = the probe list of the unsharded search
  2. per-rank ANN top-kk over the probed partitions it owns -> gather -> merge by
     (distance, rowid) keeping the OWNER of every record
  3. refine: every rank scores exactly the records it owns -> gather -> merge, keep k
  4. maximum_nprobes: queries whose MERGED ANN count is short run 1-3 again with np_max
     (the merged counts are identical on every rank, so the ranks decide together)

The shard plan and the centroid slices come from the product's own host code
(mi355_shard_plan, mi355_coarse_slice).
"""
import ctypes as C

import numpy as np

import lancedb_amd
from lancedb_amd import _abi
from lancedb_amd.distributed import coarse_slice

CAND = np.dtype([("d", "<f4"), ("pos", "<u4"), ("id", "<u8")])  # device_common.h struct Cand
EMPTY_POS = 0xFFFFFFFF


def shard_local(s, owner, rank):
    """The rows a shard handle keeps: partitions it does not own become empty."""
    po = s["part_offsets"].astype(np.int64)
    keep = [np.arange(po[p], po[p + 1]) for p in range(len(po) - 1) if owner[p] == rank]
    rows = np.concatenate(keep) if keep else np.zeros(0, np.int64)
    lens = np.array([(po[p + 1] - po[p]) if owner[p] == rank else 0 for p in range(len(po) - 1)])
    out = dict(s)
    out["part_offsets"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    out["codes"] = np.ascontiguousarray(s["codes"][rows])
    out["row_ids"] = np.ascontiguousarray(s["row_ids"][rows])
    if s.get("raw") is not None:
        out["raw"] = np.ascontiguousarray(s["raw"][rows])
    return out


class OracleShard:
    """Stands in for IvfPqIndex(shard_count=world, shard_rank=rank)."""

    def __init__(self, s, world, rank, metric="l2"):
        from oracle import oracle as orc
        self.orc = orc
        owner = lancedb_amd.shard_plan(s["part_offsets"], world)  # C ABI, host code
        loc = shard_local(s, owner, rank)
        self.ox = orc.OracleIndex(loc["centroids"], loc["codebook"], loc["part_offsets"], loc["codes"], loc["row_ids"],
                                  raw_vectors=loc.get("raw"), metric=metric)
        self.raw = loc.get("raw")
        self.rows = int(loc["part_offsets"][-1])
        self.metric = self.ox.metric
        self.nlist = self.ox.nlist
        self.scanned = 0

    def coarse_list(self, q, nprobe, lo, hi):
        """(partition id, coarse distance) records of the best min(nprobe, slice) partitions of the slice."""
        out = np.zeros(nprobe, CAND)
        out["d"], out["pos"], out["id"] = np.inf, EMPTY_POS, np.iinfo(np.uint64).max
        n_sel = min(nprobe, hi - lo)
        co = self.ox.coarse(q)[lo:hi]
        order = np.lexsort((np.arange(lo, hi), co))[:n_sel]
        out["d"][:n_sel], out["pos"][:n_sel], out["id"][:n_sel] = co[order], np.arange(n_sel), lo + order
        return out, n_sel

    def ann_list(self, q, probes, kk, params):
        """This shard's kk best (distance, local position, rowid) records over `probes`."""
        po = self.ox.part_offsets.astype(np.int64)
        cd, cp = [], []
        for p in np.asarray(probes, np.int64):
            if po[p + 1] > po[p]:
                d = self.ox.adc_partition(self.ox.build_lut(q, int(p)), int(p))
                cd.append(d)
                cp.append(np.arange(po[p], po[p + 1]))
                self.scanned += int(po[p + 1] - po[p])
        out = np.zeros(kk, CAND)
        out["d"], out["pos"], out["id"] = np.inf, EMPTY_POS, np.iinfo(np.uint64).max
        if not cd:
            return out, 0
        cd, cp = np.concatenate(cd), np.concatenate(cp)
        ok = ~np.isnan(cd)
        if params.has_lower_bound:
            ok &= cd >= np.float32(params.lower_bound)
        if params.has_upper_bound:
            ok &= cd < np.float32(params.upper_bound)
        cd, cp = cd[ok], cp[ok]
        ids = self.ox.row_ids[cp]
        order = np.lexsort((ids, cd))[:kk]
        n = len(order)
        out["d"][:n], out["pos"][:n], out["id"][:n] = cd[order], cp[order], ids[order]
        return out, n

    def exact(self, q, pos):
        L = self.orc.lib()
        qf = np.ascontiguousarray(q, np.float32)
        v = np.ascontiguousarray(self.raw[pos], np.float32)
        return np.float32(L.orc_exact_distance(qf.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p),
                                               C.c_uint32(qf.size), C.c_uint32(self.metric)))


def _slab(cands, counts):
    """[B, kk] records + [B] counts -> the byte slab ann_comm.hip gathers."""
    B = len(counts)
    cnt = np.zeros(((4 * B + 15) // 16) * 4, np.uint32)
    cnt[:B] = counts
    return np.concatenate([cands.reshape(-1).view(np.uint8), cnt.view(np.uint8), np.zeros(16, np.uint8)])


def _gather(slab, world, group=None):
    import torch  # (only the gloo model needs it: a GPU test process that has initialised RCCL through
    import torch.distributed as dist  # the C ABI must not import torch afterwards)
    mine = torch.from_numpy(slab.copy())
    out = torch.empty(world * mine.numel(), dtype=torch.uint8)
    dist.all_gather_into_tensor(out, mine, group=group)  # ONE collective per exchange
    return out.numpy().reshape(world, -1)


def _unslab(g, B, kk):
    cb = 16 * B * kk
    cands = np.stack([g[r, :cb].view(CAND).reshape(B, kk) for r in range(g.shape[0])])
    counts = np.stack([g[r, cb:cb + 4 * B].view(np.uint32) for r in range(g.shape[0])])
    return cands, counts


def _merge(cands, counts, k_out):
    """k-way merge of [world, B, kk] lists by (distance, id) -> ([B, k_out] records, owner, counts)."""
    world, B, kk = cands.shape
    out = np.zeros((B, k_out), CAND)
    out["d"], out["pos"], out["id"] = np.inf, EMPTY_POS, np.iinfo(np.uint64).max
    owner = np.full((B, k_out), 0xFFFFFFFF, np.uint32)
    cnt = np.zeros(B, np.uint32)
    for b in range(B):
        rec, own = [], []
        for r in range(world):
            c = cands[r, b, :min(int(counts[r, b]), kk)]
            c = c[(c["pos"] != EMPTY_POS) & ~np.isnan(c["d"])]
            rec.append(c)
            own.append(np.full(len(c), r, np.uint32))
        rec, own = np.concatenate(rec), np.concatenate(own)
        order = np.lexsort((rec["id"], rec["d"]))[:k_out]
        n = len(order)
        out[b, :n], owner[b, :n], cnt[b] = rec[order], own[order], n
    return out, owner, cnt


def sharded_search(shard, queries, params, world, rank, shard_coarse=False):
    """-> (rowids [B, k], distances [B, k], counts [B]) as mi355_search_sharded returns them."""
    k = params.k
    kk = k * (params.refine_factor or 1)
    nlist = shard.nlist
    np_min = min(params.nprobe_min, nlist)
    np_max = nlist if (params.nprobe_max == 0 or params.nprobe_max > nlist) else params.nprobe_max

    def ann(qs, nprobe):
        B = len(qs)
        if shard_coarse:
            lo, hi = coarse_slice(nlist, world, rank)
            lists = [shard.coarse_list(q, nprobe, lo, hi) if hi > lo else (shard.coarse_list(q, nprobe, 0, 0)[0], 0) for q in qs]
            g = _gather(_slab(np.stack([x[0] for x in lists]), np.array([x[1] for x in lists], np.uint32)), world)
            merged, _, _ = _merge(*_unslab(g, B, nprobe), nprobe)
            probes = merged["id"].astype(np.int64)
        else:
            probes = np.stack([shard.ox.select_probes(shard.ox.coarse(q), nprobe) for q in qs]).astype(np.int64)
        lists = [shard.ann_list(q, probes[i], kk, params) for i, q in enumerate(qs)]
        g = _gather(_slab(np.stack([x[0] for x in lists]), np.array([x[1] for x in lists], np.uint32)), world)
        return _merge(*_unslab(g, B, kk), kk)

    def finish(qs, glist, gowner, gcnt):
        B = len(qs)
        if not params.refine_factor:
            return glist["id"][:, :k].copy(), glist["d"][:, :k].copy(), np.minimum(gcnt, k)
        mine = np.zeros((B, kk), CAND)
        mine["d"], mine["pos"], mine["id"] = np.inf, EMPTY_POS, np.iinfo(np.uint64).max
        for b in range(B):
            for c in range(int(gcnt[b])):
                if gowner[b, c] != rank:
                    continue
                d = shard.exact(qs[b], int(glist[b, c]["pos"]))
                if np.isnan(d) or (params.has_lower_bound and not d >= np.float32(params.lower_bound)) or \
                        (params.has_upper_bound and not d < np.float32(params.upper_bound)):
                    continue
                mine[b, c] = (d, glist[b, c]["pos"], glist[b, c]["id"])
        g = _gather(_slab(mine, gcnt), world)
        out, _, cnt = _merge(*_unslab(g, B, kk), k)
        return out["id"], out["d"], cnt

    qs = np.ascontiguousarray(queries, np.float32)
    glist, gowner, gcnt = ann(qs, np_min)
    ids, d, cnt = finish(qs, glist, gowner, gcnt)
    if np_max > np_min:
        short = np.nonzero(gcnt < kk)[0]  # the same on every rank
        if len(short):
            g2, o2, c2 = ann(qs[short], np_max)
            i2, d2, n2 = finish(qs[short], g2, o2, c2)
            ids[short], d[short], cnt[short] = i2, d2, n2
    return ids, d, cnt
