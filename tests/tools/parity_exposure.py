"""How exposed is "bit-exact row ids" to the summation order nobody can pin here?  (VERDICT r2 weak #1.)

The engine and the oracle share ONE arithmetic contract (oracle/ann_oracle.c: d-ascending fmaf chains, ADC in
sub-quantiser order).  lance-linalg's kernels are not in the container; on AVX2 they most likely keep several lane
accumulators, do not fuse, and reduce with a tree.  This tool measures, on the trained recall index of bench.py
(2 M x 768 mixture, nlist 1024, m = 96), what fraction of queries' top-10 would CHANGE if the distances were
computed (a) in float64, (b) in a plausible non-fused multi-lane f32 order — i.e. the size of the claim a maintainer
takes on trust when reading "bit-exact".  It also re-derives the contract's values in numpy and checks them against
the engine's returned distances bit for bit (so the re-implementation the study rests on is itself verified).

usage (GPU box): python tests/tools/parity_exposure.py [queries] > profiles/r03_parity_exposure.json"""
import json
import sys
import time
import types

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402

NQ = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
a = types.SimpleNamespace(recall_rows=2_000_000, recall_queries=NQ, recall_iters=25)
dim, m = 768, 96
dsub = dim // m
t0 = time.perf_counter()
R = bench.recall_index(a, dim, m)
import torch  # noqa: E402
import lancedb_amd  # noqa: E402

ix = lancedb_amd.IvfPqIndex(R["cen"].contiguous(), R["codebook"].contiguous(), R["part_offsets"], R["codes"], R["order"], raw_vectors=R["xs"].float())
q = R["q"].cpu().numpy()
cen = R["cen"].cpu().numpy()
cb = R["codebook"].cpu().numpy()            # [m, 256, dsub]
po = np.asarray(R["part_offsets"]).astype(np.int64)
codes = R["codes"].cpu().numpy().reshape(-1, m)  # index order
order = R["order"].cpu().numpy().astype(np.int64)  # index position -> row id
xs = R["xs"].float().cpu().numpy()                  # raw rows in index order
pos_of = np.empty(len(order), np.int64)
pos_of[order] = np.arange(len(order))


def fma32(x, y, z):  # float32 fmaf through float64 (exact: 53 >= 2 * 24 + 2)
    return (x.astype(np.float64) * y.astype(np.float64) + z.astype(np.float64)).astype(np.float32)


def lut_terms(qv, ids):
    """per candidate row: the m squared-distance terms |r_j - e_j|^2 in three arithmetics"""
    pos = pos_of[ids]
    part = np.searchsorted(po, pos, side="right") - 1
    r = (qv[None, :] - cen[part]).reshape(len(ids), m, dsub)          # f32, exact op order of the contract
    e = cb[np.arange(m)[None, :], codes[pos]]                          # [n, m, dsub]
    d = r - e                                                          # f32
    # contract: d-ascending fmaf chain
    acc = np.zeros(d.shape[:2], np.float32)
    for t in range(dsub):
        acc = fma32(d[..., t], d[..., t], acc)
    # non-fused, one 8-lane vector per sub-vector, tree reduction (the likely AVX2 shape)
    sq = d * d                                                         # f32 products, rounded
    tree = ((sq[..., 0] + sq[..., 1]) + (sq[..., 2] + sq[..., 3])) + ((sq[..., 4] + sq[..., 5]) + (sq[..., 6] + sq[..., 7]))
    # float64 from the f32 inputs
    r64 = (qv[None, :].astype(np.float64) - cen[part].astype(np.float64)).reshape(len(ids), m, dsub)
    ex = ((r64 - e.astype(np.float64)) ** 2).sum(-1)
    return acc, tree, ex


def adc_f32(terms):  # sub-quantiser order, plain f32 adds (the contract's and, by its transposed layout, lance's order)
    out = np.zeros(terms.shape[0], np.float32)
    for j in range(m):
        out = out + terms[:, j]
    return out


def exact_variants(qv, rows):
    v = xs[rows]
    t = qv[None, :] - v                                                # f32
    acc = np.zeros(len(rows), np.float32)
    for dd in range(dim):
        acc = fma32(t[:, dd], t[:, dd], acc)                           # contract chain
    sq = (t * t).reshape(len(rows), dim // 16, 16)
    lanes = np.zeros((len(rows), 16), np.float32)
    for c in range(dim // 16):
        lanes = lanes + sq[:, c]                                       # 16 lane accumulators, non-fused
    w = lanes
    while w.shape[1] > 1:
        w = w[:, 0::2] + w[:, 1::2]                                    # tree reduction
    ex = ((qv[None, :].astype(np.float64) - v.astype(np.float64)) ** 2).sum(1)
    return acc, w[:, 0], ex


def top10(ids, d):
    o = np.lexsort((ids, d))[:10]
    return ids[o]


ann = ix.search(q, k=50, nprobe_min=64, nprobe_max=64)
ref = ix.search(q, k=10, nprobe_min=64, nprobe_max=64, refine_factor=25)
pool = ix.search(q, k=250, nprobe_min=64, nprobe_max=64)  # the candidates refine re-ranks
res = {"index": "bench.py recall index: 2 M x 768 Gaussian mixture, nlist 1024, m 96, trained + encoded by the engine",
       "queries": NQ, "nprobe": 64}
c = {k: 0 for k in ("adc_contract_equals_engine", "adc_set_tree", "adc_order_tree", "adc_set_f64", "adc_order_f64",
                    "ref_contract_equals_engine", "ref_set_lanes", "ref_order_lanes", "ref_set_f64", "ref_order_f64")}
rel_adc = rel_ref = 0.0
gap_adc = []
for i in range(NQ):
    ids = ann.rowids[i].astype(np.int64)
    acc, tree, ex = lut_terms(q[i], ids)
    d_con, d_tree, d_ex = adc_f32(acc), adc_f32(tree), ex.sum(1)
    c["adc_contract_equals_engine"] += int((d_con == ann.distances[i]).all())
    base = top10(ids, d_con)
    for name, d in (("tree", d_tree), ("f64", d_ex)):
        t = top10(ids, d)
        c["adc_set_" + name] += int(set(t.tolist()) != set(base.tolist()))
        c["adc_order_" + name] += int((t != base).any())
    rel_adc = max(rel_adc, float(np.max(np.abs(d_con.astype(np.float64) - d_ex) / d_ex)))
    s10 = np.sort(d_con)
    gap_adc.append(float((s10[10] - s10[9]) / s10[9]))
    rows = pos_of[pool.rowids[i].astype(np.int64)]
    rid = pool.rowids[i].astype(np.int64)
    e_con, e_lane, e_ex = exact_variants(q[i], rows)
    base = top10(rid, e_con)
    c["ref_contract_equals_engine"] += int((base == ref.rowids[i].astype(np.int64)).all() and (np.sort(e_con)[:10] == ref.distances[i]).all())
    for name, d in (("lanes", e_lane), ("f64", e_ex)):
        t = top10(rid, d)
        c["ref_set_" + name] += int(set(t.tolist()) != set(base.tolist()))
        c["ref_order_" + name] += int((t != base).any())
    rel_ref = max(rel_ref, float(np.max(np.abs(e_con.astype(np.float64) - e_ex) / e_ex)))
res["adc"] = {
    "numpy_contract_equals_engine_distances_queries": c["adc_contract_equals_engine"],
    "top10_SET_changes_under_tree_summed_lut": c["adc_set_tree"] / NQ, "top10_ORDER_changes_under_tree_summed_lut": c["adc_order_tree"] / NQ,
    "top10_SET_changes_under_float64": c["adc_set_f64"] / NQ, "top10_ORDER_changes_under_float64": c["adc_order_f64"] / NQ,
    "max_rel_error_contract_vs_float64": rel_adc,
    "relative_gap_rank10_to_rank11_median": float(np.median(gap_adc)), "relative_gap_p01": float(np.percentile(gap_adc, 1))}
res["refine_rf25"] = {
    "numpy_contract_equals_engine_queries": c["ref_contract_equals_engine"],
    "top10_SET_changes_under_16_lane_nonfused": c["ref_set_lanes"] / NQ, "top10_ORDER_changes_under_16_lane_nonfused": c["ref_order_lanes"] / NQ,
    "top10_SET_changes_under_float64": c["ref_set_f64"] / NQ, "top10_ORDER_changes_under_float64": c["ref_order_f64"] / NQ,
    "max_rel_error_contract_vs_float64": rel_ref}
res["reading"] = ("SET change = a different row id in the returned top-10; ORDER change = same ids, another order or a different id. "
                  "Candidate pools: the engine's ANN top-50 (ADC) and top-250 (refine), so flips at the rank-10 boundary are all seen. "
                  "The summation orders are models of what an AVX2 build of lance-linalg plausibly does ([EXT], not verified).")
res["seconds"] = round(time.perf_counter() - t0, 1)
print(json.dumps(res, indent=1))
