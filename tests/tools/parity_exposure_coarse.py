"""How exposed are the returned row ids to the FORM of the coarse distance?  (VERDICT r5 weak #1b.)

The contract (oracle/ann_oracle.c, every kernel) scores a centroid with the expanded form fmaf(-2, q.c, |q|^2 + |c|^2), both
dot products d-ascending fmaf chains — the form that is a GEMM.  lance's partition finder most likely computes l2(q, c)
directly, with SIMD lane accumulators ([EXT], lance-linalg is not in the container).  The two forms round differently, so the
nprobes-th and (nprobes + 1)-th nearest partitions can swap: a different partition is scanned, and the top-10 MAY change.
On the trained index of the bench's recall leg (10 M x 768, nlist 4096, m 96; the embedding-like column or rounds 1-5's
Gaussian mixture) this tool measures, at
nprobes 20 and 64:
  * how many queries get a different probe SET under (a) a direct, non-fused, 16-lane-summed f32 l2 and (b) float64;
  * how many of those queries then return a different top-10 (ids), by scanning the alternative probe list with the engine
    (mi355_search_probes) — the pool the exposure is measured on is complete: a changed probe list is actually searched.
The engine's own probe list is taken from mi355_coarse_topn and cross-checked against a numpy restatement of the contract
on a sample.

usage (GPU box): python tests/tools/parity_exposure_coarse.py [queries] [rows] [embedding|mixture] > profiles/r06_parity_exposure.json"""
import json
import os
import sys
import time
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

NQ = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ROWS = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
DATA = sys.argv[3] if len(sys.argv) > 3 else "embedding"  # bench.recall_column: "embedding" (the bench's recall leg) | "mixture" (rounds 1-5)
a = types.SimpleNamespace(recall_rows=ROWS, recall_queries=NQ, recall_iters=25)
dim, m = 768, 96
t0 = time.perf_counter()
R = bench.recall_index(a, dim, m, data=DATA)
import torch  # noqa: E402
import lancedb_amd  # noqa: E402

del R["x"]
torch.cuda.empty_cache()
ix = lancedb_amd.IvfPqIndex(R["cen"].contiguous(), R["codebook"].contiguous(), R["part_offsets"], R["codes"], R["order"])
q = R["q"].cpu().numpy()
cen = R["cen"].cpu().numpy()
nlist = cen.shape[0]


def fma32(x, y, z):  # float32 fmaf through float64 (exact: 53 >= 2 * 24 + 2)
    return (x.astype(np.float64) * y.astype(np.float64) + z.astype(np.float64)).astype(np.float32)


def chain_dot(x, y):  # d-ascending fmaf chain over the last axis, f32
    acc = np.zeros(x.shape[:-1], np.float32)
    for t in range(x.shape[-1]):
        acc = fma32(x[..., t], y[..., t], acc)
    return acc


cc = chain_dot(cen, cen)  # |c|^2, the handle's cnorm


def scores(qv):
    """the three arithmetics of d(q, c) for all centroids"""
    qq = chain_dot(qv[None, :], qv[None, :])[0]
    contract = fma32(np.float32(-2.0) * np.ones(nlist, np.float32), chain_dot(np.broadcast_to(qv, cen.shape), cen), qq + cc)
    t = qv[None, :] - cen                                   # f32
    sq = (t * t).reshape(nlist, dim // 16, 16)              # non-fused products
    lanes = np.zeros((nlist, 16), np.float32)
    for c in range(dim // 16):
        lanes = lanes + sq[:, c]                            # 16 lane accumulators
    w = lanes
    while w.shape[1] > 1:
        w = w[:, 0::2] + w[:, 1::2]                         # tree reduction
    f64 = ((qv[None, :].astype(np.float64) - cen.astype(np.float64)) ** 2).sum(1)
    return contract, w[:, 0], f64


def probes_of(score, nprobe):
    return np.lexsort((np.arange(nlist), score))[:nprobe]


res = {"index": f"bench.py recall index: {ROWS} x 768, {R['data']}, nlist {nlist}, m 96, trained + encoded by the engine",
       "queries": NQ,
       "forms": {"contract": "fmaf(-2, q.c, |q|^2 + |c|^2), d-ascending fmaf chains (oracle/ann_oracle.c; every kernel)",
                 "direct_16_lanes": "sum (q - c)^2, products not fused, 16 f32 lane accumulators, tree reduction ([EXT] model of an AVX2 build)",
                 "float64": "sum (q - c)^2 in float64 from the f32 inputs"}}
sc = [scores(q[i]) for i in range(NQ)]
for nprobe in (20, 64):
    eng_p, _, _ = ix.coarse_topn(q, nprobe)
    eng_p = np.asarray(eng_p).astype(np.int64)
    base = ix.search(q, k=10, nprobe_min=nprobe, nprobe_max=nprobe)
    out = {"numpy_contract_equals_engine_probe_sets": int(sum(set(probes_of(sc[i][0], nprobe).tolist()) == set(eng_p[i].tolist()) for i in range(NQ)))}
    for name, col in (("direct_16_lanes", 1), ("float64", 2)):
        alt = np.stack([probes_of(sc[i][col], nprobe) for i in range(NQ)]).astype(np.uint64)
        changed = np.array([set(alt[i].tolist()) != set(eng_p[i].tolist()) for i in range(NQ)])
        got = ix.search_probes(q, alt, k=10)
        ids_differ = np.array([set(got.rowids[i][:got.counts[i]].tolist()) != set(base.rowids[i][:base.counts[i]].tolist()) for i in range(NQ)])
        order_differ = np.array([(got.rowids[i] != base.rowids[i]).any() for i in range(NQ)])
        # the margin at the cut: relative gap between the nprobe-th and (nprobe + 1)-th score, contract arithmetic
        gaps = []
        for i in range(NQ):
            s = np.sort(sc[i][0])
            gaps.append(float((s[nprobe] - s[nprobe - 1]) / s[nprobe - 1]))
        out[name] = {"probe_SET_changes": float(changed.mean()), "top10_SET_changes": float(ids_differ.mean()),
                     "top10_ORDER_changes": float(order_differ.mean()),
                     "top10_SET_changes_among_changed_probe_sets": float(ids_differ[changed].mean()) if changed.any() else 0.0,
                     "queries_with_changed_probe_set": int(changed.sum())}
        out["relative_gap_at_the_cut_median"] = float(np.median(gaps))
        out["relative_gap_at_the_cut_p01"] = float(np.percentile(gaps, 1))
    res[f"nprobes_{nprobe}"] = out
res["reading"] = ("probe_SET_changes = the query scans at least one different partition; top10_SET_changes = it then returns at least one "
                  "different row id (measured by scanning the alternative probe list with the engine).  A swapped partition is the one at the "
                  "cut — the farthest probed one — so it rarely holds a top-10 row.  The lane model is a guess at lance-linalg's kernel "
                  "([EXT], not verified); float64 bounds what ANY f32 arithmetic could differ by.")
res["seconds"] = round(time.perf_counter() - t0, 1)
print(json.dumps(res, indent=1))
