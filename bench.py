#!/usr/bin/env python3
"""bench.py — queries/sec of the IVF-PQ search hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N
            --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

Workload (BASELINE.json `metric`, configs[2] = C3): IVF-PQ 100 M x 768, nlist
4096, PQ m = 96 x 8 bit, nprobe 64, k 10, L2.  A *step* is one pass of the whole
search path (coarse quantiser -> probe select -> LUT build + ADC scan + top-k ->
merge) over one batch of `--batch` queries whose vectors are already resident
in HBM; results stay in HBM.  value = queries / second over all N GPUs.

Synthetic data: the index cannot be trained from 307 GB of raw
vectors, so centroids ~ N(0,1), codebook ~ N(0,0.25), uniform u8 codes,
log-normally skewed partition lengths (sigma 0.5) and a random row-id
permutation are generated on the device with seed 0x1A2CE; queries are
centroid[random] + N(0, 0.25).  PyTorch is used only for device memory, the RNG
and torch.distributed; every timed kernel is the engine's own HIP code behind
the C ABI.

N > 1 shards the IVF partition list (greedy bytes-balanced plan) with the coarse
quantiser replicated; per step every rank scans the probed partitions it owns,
then ONE all-gather of the per-shard top-k candidates (RCCL over xGMI) and a
k-way merge on every rank.  The index size is fixed, so scaling is "strong".

One JSON line on rank 0, with `roofline` (dominant kernel = the ADC scan, HIP
events recorded on the search stream inside the timed region) and, at N = 1,
`cpu_baseline` (the C oracle on a bounded sample of the same queries, also used
as a full-size parity check).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 0x1A2CE
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=2048, help="queries per step")
    ap.add_argument("--n-rows", type=int, default=100_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--m", type=int, default=96)
    ap.add_argument("--nprobe", type=int, default=64)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--skew", type=float, default=0.5, help="sigma of the log-normal partition-length skew")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU-oracle baseline (0 = skip)")
    ap.add_argument("--recall-rows", type=int, default=500_000,
                    help="rows of the TRAINED index recall@10 is measured on (0 = skip); the 100 M throughput "
                         "index has random codes, so recall is only meaningful on a trained one")
    ap.add_argument("--workload", default="ivfpq", choices=["ivfpq", "flat"],
                    help="ivfpq = C3, the configuration BASELINE.json's metric is quoted on (default); "
                         "flat = C2 (10 M x 768 bf16, 1024 queries), a secondary line for the MFMA path")
    ap.add_argument("--flat-rows", type=int, default=10_000_000)
    ap.add_argument("--flat-batch", type=int, default=1024)
    ap.add_argument("--flat-metric", default="l2", choices=["l2", "cosine", "dot"])
    ap.add_argument("--flat-gemm", type=int, default=0, help="mi355_flat_configure gemm_variant (0 = the library's choice)")
    ap.add_argument("--flat-grid", type=int, default=0, help="mi355_flat_configure grid_workgroups")
    ap.add_argument("--scan-variant", type=int, default=0)
    ap.add_argument("--slice-rows", type=int, default=0)
    return ap.parse_args()


def main_flat(a):
    """C2: flat L2 / cosine over a bf16 column as an MFMA GEMM filter + exact re-rank.
    Single GPU (rows would shard the same way as IVF partitions; not wired up)."""
    import numpy as np
    import torch

    import lancedb_amd
    from lancedb_amd import _abi
    if a.gpus != 1:
        raise SystemExit("--workload flat is a single-GPU line")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n, dim, B, k = a.flat_rows, a.dim, a.flat_batch, a.k
    g = torch.Generator(device=dev)
    g.manual_seed(SEED)
    col = torch.randn((n, dim), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    P = 2
    qpool = [torch.randn((B, dim), generator=g, device=dev, dtype=torch.float32) for _ in range(P)]
    torch.cuda.synchronize()
    fl = lancedb_amd.FlatIndex(col.view(torch.int16), dtype=_abi.DTYPE_BF16, device=0)
    stream = torch.cuda.current_stream().cuda_stream
    fl.set_stream(stream)
    fl.configure(gemm_variant=a.flat_gemm, grid_workgroups=a.flat_grid)
    mt = _abi.METRIC_NAMES[a.flat_metric]
    params = _abi.make_params(k=k, nprobe_min=1, nprobe_max=1, metric=mt)
    out = (torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
           torch.empty((B,), dtype=torch.int32, device=dev))
    for i in range(a.warmup):
        fl.search(qpool[i % P], params, out=out)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(a.steps):
        last = fl.search(qpool[i % P], params, out=out)
    ev[1].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert fl.info() == (1, 1), "the MFMA filter path did not run"
    flops = 2.0 * B * n * dim
    ms = elapsed / a.steps * 1e3
    achieved = flops / (elapsed / a.steps) / 1e12
    result = {
        "metric": "queries/sec, flat (no index) KNN 10M×768 bf16, batch 1024, k=10 (BASELINE.json configs[1])",
        "value": B * a.steps / elapsed, "unit": "queries/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic",
        "config": {"workload": f"flat_{n}x{dim}_bf16_batch{B}_k{k}_{a.flat_metric}", "n_rows": n, "dim": dim,
                   "batch_queries": B, "k": k},
        "roofline": {"bound": "mfma", "kernel": "whole step (k_flat_gemm dominates; + segmin/tau/compact/rerank)",
                     "achieved": achieved, "peak": 2500.0, "unit": "TFLOP/s", "frac": achieved / 2500.0,
                     "traffic": traffic_from_profiles(f"flat_{n}x{dim}_bf16_batch{B}_k{k}_{a.flat_metric}", B),
                     "algorithmic_flops_per_step": flops},
    }
    if a.cpu_seconds > 0:
        from oracle import oracle as orc
        orc.build()
        cores = os.cpu_count() or 1
        nq = min(B, 32)  # every query sweeps the whole column on the host: keep the sample to ~15 s
        hv = col.view(torch.int16).cpu().numpy().view(np.uint16)
        hq = qpool[(a.steps - 1) % P][:nq].cpu().numpy()
        t1 = time.perf_counter()
        ids, dist, cnt, _ = orc.flat_search(hv, hq, k=k, dtype=_abi.DTYPE_BF16, metric=mt)
        dt = time.perf_counter() - t1
        g_ids = last.rowids[:nq].cpu().numpy().astype(np.uint64)
        g_dist = last.distances[:nq].cpu().numpy()
        result["cpu_baseline"] = {
            "value": nq / dt, "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"{nq} queries of the last timed batch, one per thread, {dt:.1f} s on {cores} host cores; "
                      "C restatement (oracle/ann_oracle.c), not the reference binary",
            "parity": {"queries": nq, "rowids_bit_exact": bool((g_ids == ids).all()),
                       "distances_equal": bool((g_dist == dist).all())}}
    print(json.dumps(result), flush=True)


def main():
    a = parse()
    if a.workload == "flat":
        return main_flat(a)
    import numpy as np
    import torch
    import torch.distributed as dist

    import lancedb_amd
    from lancedb_amd import _abi

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    n, dim, nlist, m = a.n_rows, a.dim, a.nlist, a.m
    dsub = dim // m
    # ---- synthetic index, identical on every rank (same seed, same device type)
    g = torch.Generator(device=dev)
    g.manual_seed(SEED)
    centroids = torch.randn((nlist, dim), generator=g, device=dev, dtype=torch.float32)
    codebook = torch.randn((m, 256, dsub), generator=g, device=dev, dtype=torch.float32) * 0.5
    rng = np.random.default_rng(SEED)
    w = np.exp(rng.normal(0.0, a.skew, size=nlist))
    lens = rng.multinomial(n, w / w.sum())
    part_offsets = np.zeros(nlist + 1, dtype=np.uint64)
    part_offsets[1:] = np.cumsum(lens)
    codes = torch.randint(0, 256, (n * m,), generator=g, device=dev, dtype=torch.uint8)
    row_ids = torch.randperm(n, generator=g, device=dev)  # int64, a permutation of 0..n
    torch.cuda.synchronize()

    t_open = time.time()
    # the uniform code bytes are declared to be in lance's per-partition transposed layout
    ix = lancedb_amd.IvfPqIndex(centroids, codebook, part_offsets, codes, row_ids, metric="l2",
                                codes_layout=_abi.CODES_PART_TRANSPOSED, device=local_rank,
                                shard_count=world, shard_rank=rank)
    t_open = time.time() - t_open
    rows_local, parts_local = ix.info()

    want_cpu = rank == 0 and world == 1 and a.cpu_seconds > 0
    h_codes = h_rowids = None
    if want_cpu:
        h_codes = codes.cpu().numpy()
        h_rowids = row_ids.cpu().numpy().astype(np.uint64, copy=False)
    h_centroids = centroids.cpu().numpy()
    h_codebook = codebook.cpu().numpy()
    del codes, row_ids
    torch.cuda.empty_cache()

    # ---- query batches (resident in HBM before the timed region)
    P = 4
    qpool = []
    for _ in range(P):
        pick = torch.randint(0, nlist, (a.batch,), generator=g, device=dev)
        qpool.append((centroids[pick] + 0.5 * torch.randn((a.batch, dim), generator=g, device=dev)).contiguous())
    params = _abi.make_params(k=a.k, nprobe_min=a.nprobe, nprobe_max=a.nprobe)
    B, k = a.batch, a.k
    out = (torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
           torch.empty((B,), dtype=torch.int32, device=dev))
    stream = torch.cuda.current_stream().cuda_stream
    ix.set_stream(stream)  # engine kernels, RCCL and torch share one ordered stream
    ix.configure(scan_variant=a.scan_variant, slice_rows=a.slice_rows, profile=0)
    comm = searcher = None
    if world > 1:
        # the exchange is RCCL behind the C ABI (mi355_comm_* / mi355_search_sharded: one packed
        # all-gather of the per-shard candidate records on the search stream + a k-way merge on every
        # rank); torch.distributed only carries the 128-byte communicator id to the other ranks
        from lancedb_amd.distributed import Comm, ShardedSearcher, unique_id
        uid = [unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        comm = Comm(uid[0], rank, world, device=local_rank)
        searcher = ShardedSearcher(ix, comm)

    def step(i):
        r = searcher.search(qpool[i % P], params, out=out) if searcher else ix.search(qpool[i % P], params, out=out)
        return r.rowids, r.distances, r.counts

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    fence()
    ix.configure(scan_variant=a.scan_variant, slice_rows=a.slice_rows, profile=2)  # cumulative, non-blocking
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        last = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = ix.stats()

    # roofline of the dominant kernel (ADC scan): algorithmic code bytes per launch
    launches = max(st["scan_launches"], 1)
    bytes_per_launch = st["code_bytes_scanned"] / launches
    us_per_launch = st["us_scan"] / launches
    stat = torch.tensor([st["code_bytes_scanned"], st["us_scan"], float(launches)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(stat, op=dist.ReduceOp.SUM)  # whole-job bytes / summed kernel time
    achieved = (float(stat[0]) / float(stat[2])) / (float(stat[1]) / float(stat[2]) * 1e-6) / 1e9 if float(stat[1]) > 0 else 0.0
    qps = a.batch * a.steps / elapsed
    workload = f"ivfpq_{n}x{dim}_nlist{nlist}_m{m}x8_nprobe{a.nprobe}_k{a.k}_l2"
    traffic = traffic_from_profiles(workload, a.batch) if world == 1 else None

    result = {
        "metric": "queries/sec @ recall@10, 100M×768 IVF-PQ nprobe=64 k=10",
        "value": qps,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": workload,
            "batch_queries": a.batch, "n_rows": n, "dim": dim, "nlist": nlist, "m": m, "nprobe": a.nprobe,
            "k": a.k, "partition_skew_sigma": a.skew, "parallelism": f"ivf_partition_shard{world}",
            "scan_variant": st["scan_variant"], "rows_on_rank0": rows_local, "partitions_on_rank0": parts_local,
            "index_open_s": round(t_open, 2),
        },
        "roofline": {
            "bound": "hbm", "kernel": "k_scan (LUT build + ADC scan + top-k)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "algorithmic_bytes_per_launch": bytes_per_launch, "us_per_launch": us_per_launch,
            "launches": int(float(stat[2])),
            "stage_us_per_step": {s: st["us_" + s] / a.steps for s in ("coarse", "select", "scan", "merge")},
        },
    }

    if rank == 0 and world == 1 and a.recall_rows > 0:
        result["recall_at_10"] = recall_at_10(a, np, dim, m)
    if want_cpu:
        result["cpu_baseline"] = cpu_baseline(a, np, h_centroids, h_codebook, part_offsets, h_codes, h_rowids,
                                              qpool[(a.steps - 1) % P], last, params)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def recall_at_10(a, np, dim, m):
    """recall@10 of the engine's IVF-PQ search against exact flat search (the
    engine's own flat path, bit-identical to the oracle's exact sweep) on a REAL
    index: Gaussian-mixture vectors; IVF centroids and residual PQ codebooks
    trained and all rows encoded by the engine's own build entry points (index
    parameters follow the reference's builder, 8-bit PQ with m sub-vectors,
    rust/lancedb/src/index/vector.rs:61-119, :266-319).  The 100 M throughput index has
    random codes, so recall is only meaningful here."""
    import torch
    import lancedb_amd
    t0 = time.perf_counter()
    n, nlist, nq, dsub = a.recall_rows, 1024, 1000, dim // m
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(SEED)
    comps = torch.randn((2048, dim), generator=g, device=dev) * 1.5
    x = comps[torch.randint(0, 2048, (n,), generator=g, device=dev)] + torch.randn((n, dim), generator=g, device=dev)
    q = comps[torch.randint(0, 2048, (nq,), generator=g, device=dev)] + torch.randn((nq, dim), generator=g, device=dev)
    # the engine's own build path (mi355_kmeans_train / mi355_ivf_residuals / mi355_ivfpq_encode),
    # device-resident; sample sizes as the reference's sample_rate = 256 (index/vector.rs:76-91)
    iters = 6
    torch.cuda.synchronize()
    pick = torch.randperm(n, generator=g, device=dev)
    ivf_rows = x[pick[:min(n, 256 * nlist)].sort().values].contiguous()
    init = ivf_rows[torch.randperm(ivf_rows.shape[0], generator=g, device=dev)[:nlist].sort().values].contiguous()
    torch.cuda.synchronize()
    t_train = time.perf_counter()
    cen, _ = lancedb_amd.kmeans_train(ivf_rows, init, iters=iters)
    pq_rows = x[pick[:min(n, 256 * 256)].sort().values].contiguous()
    torch.cuda.synchronize()
    resid, _ = lancedb_amd.ivf_residuals(pq_rows, cen)
    seeds = resid[torch.randperm(resid.shape[0], generator=g, device=dev)[:256].sort().values]
    codebook = torch.empty((m, 256, dsub), device=dev)
    for j in range(m):
        cb0 = seeds[:, j * dsub:(j + 1) * dsub].contiguous()
        torch.cuda.synchronize()
        cb, _ = lancedb_amd.kmeans_train(resid, cb0, iters=iters, cols=(j * dsub, (j + 1) * dsub))
        codebook[j] = cb
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t_train
    t_enc = time.perf_counter()
    part_offsets, codes, order = lancedb_amd.ivfpq_encode(x, cen, codebook)
    t_enc = time.perf_counter() - t_enc
    xs = x[order].contiguous()
    torch.cuda.synchronize()
    ix = lancedb_amd.IvfPqIndex(cen.contiguous(), codebook.contiguous(), part_offsets, codes, order, raw_vectors=xs)
    fl = lancedb_amd.FlatIndex(x.contiguous())
    torch.cuda.synchronize()
    hq = q.cpu().numpy()
    truth = fl.search(hq, k=10).rowids
    out = {"n_rows": n, "dim": dim, "nlist": nlist, "m": m, "queries": nq, "truth": "exact flat search (engine flat path)",
           "data": "2048-component Gaussian mixture; IVF + residual PQ trained by 6 Lloyd iterations "
                   "(mi355_kmeans_train), rows encoded by mi355_ivfpq_encode",
           "train_seconds": round(t_train, 2), "encode_rows_per_s": round(n / t_enc)}
    for nprobe, rf in ((64, 0), (64, 10), (16, 0)):
        got = ix.search(hq, k=10, nprobe_min=nprobe, nprobe_max=nprobe, refine_factor=rf).rowids
        rec = float(np.mean([len(set(truth[i].tolist()) & set(got[i].tolist())) / 10.0 for i in range(nq)]))
        out[f"nprobe{nprobe}" + (f"_refine{rf}" if rf else "")] = round(rec, 4)
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


def traffic_from_profiles(workload, batch):
    """HBM bytes per launch of the scan kernel.  PMC counters cannot be read from
    inside the timed process (rocprofv3 collects them in a separate pass, which
    must not be combined with timing), so this is the committed measurement of
    the SAME workload under profiles/ (FETCH_SIZE x 2, the gfx950 correction of
    MI355X_MICROARCH.md); null when no matching measurement is committed."""
    import glob
    best = None
    paths = glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")) + \
        glob.glob(os.path.join(ROOT, "profiles", "*fetch_size*.json"))
    for path in sorted(paths):
        try:
            t = json.load(open(path))
        except (OSError, ValueError):
            continue
        if t.get("workload") == workload and t.get("batch_queries") == batch:
            best = t.get("traffic_bytes_per_launch")
    return best


def cpu_baseline(a, np, centroids, codebook, part_offsets, codes, row_ids, d_queries, last, params):
    """The C oracle (a restatement, kind = "port") on this box's host cores over a
    bounded sample of the last batch; doubles as the full-size parity check."""
    from oracle import oracle as orc
    orc.build()
    cores = os.cpu_count() or 1
    ox = orc.OracleIndex(centroids, codebook, part_offsets, codes, row_ids, metric="l2",
                         codes_layout=1, borrow=True)
    q = d_queries.cpu().numpy()
    t0 = time.perf_counter()
    ids0, dist0, cnt0, st = ox.search(q[:cores], params)
    t_probe = time.perf_counter() - t0
    n_more = int(max(0, min(len(q) - cores, (a.cpu_seconds - t_probe) / max(t_probe, 1e-3) * cores)))
    n_more -= n_more % cores
    t1 = time.perf_counter()
    if n_more:
        ids1, dist1, cnt1, _ = ox.search(q[cores:cores + n_more], params)
    t_more = time.perf_counter() - t1
    nq = cores + n_more
    ids = np.concatenate([ids0, ids1]) if n_more else ids0
    dst = np.concatenate([dist0, dist1]) if n_more else dist0
    g_ids = last[0][:nq].cpu().numpy().astype(np.uint64)
    g_dist = last[1][:nq].cpu().numpy()
    rel = float(np.max(np.abs(g_dist - dst) / np.maximum(np.abs(dst), 1e-30)))
    return {
        "value": nq / (t_probe + t_more), "unit": "queries/s", "cores": cores, "kind": "port",
        "sample": f"{nq} queries of the last timed batch, one query per thread (OpenMP), "
                  f"{t_probe + t_more:.1f} s of wall time on {cores} host cores; C restatement of the "
                  "lance-index IVF-PQ path (oracle/ann_oracle.c, -O3 -mavx2 -mfma), not the reference binary",
        "parity": {"queries": nq, "rowids_bit_exact": bool((g_ids == ids).all()), "max_rel_distance_error": rel},
    }


if __name__ == "__main__":
    main()
