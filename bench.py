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
then ONE packed all-gather of the per-shard top-k candidate records (RCCL over
xGMI, behind the C ABI: mi355_search_sharded) and a k-way merge on every rank.
Every rank generates ONLY the partitions it owns (per-partition seeds), so the
index is the same for every N.  The index size is fixed: scaling is "strong".

One JSON line on rank 0, with `roofline` (dominant kernel = the ADC scan, HIP
events recorded on the search stream inside the timed region) and, at N = 1,
`cpu_baseline` (the C oracle on a bounded sample of the same queries, also used
as a full-size parity check), `recall_at_10` (a trained 10 M-row index with its bf16 raw column in HBM: QPS AND recall@10 of
every operating point on that one index, engine and oracle; `qps_vs_recall_at_10` lists them), and `secondary`: single-query latency and 64-thread throughput of host callers,
the same 100 M index with refine_factor 10 / 25 over resident bf16 raw vectors, and flat C2
(BASELINE.json configs[1]: 10 M x 768 bf16, 1024 queries, L2 and cosine) with the
GEMM kernel's own roofline and a full-size parity check each.
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
    ap.add_argument("--recall-rows", type=int, default=10_000_000,
                    help="rows of the TRAINED index recall@10 is measured on (0 = skip); the 100 M throughput "
                         "index has random codes, so recall is only meaningful on a trained one")
    ap.add_argument("--recall-queries", type=int, default=10_000)
    ap.add_argument("--recall2-rows", type=int, default=1_000_000,
                    help="rows of the second, embedding-like recall set (1536-d unit vectors of low intrinsic dimension, cosine; 0 = skip)")
    ap.add_argument("--recall2-queries", type=int, default=2_000)
    ap.add_argument("--recall-iters", type=int, default=25, help="Lloyd iterations of the IVF and PQ trainers")
    ap.add_argument("--secondary", type=int, default=1, help="0 = skip the secondary lines (refine operating point, flat C2)")
    ap.add_argument("--workload", default="ivfpq", choices=["ivfpq", "flat", "c4"],
                    help="ivfpq = C3, the configuration BASELINE.json's metric is quoted on (default); "
                         "flat = C2 (10 M x 768 bf16, 1024 queries), a secondary line for the MFMA path; "
                         "c4 = BASELINE.json configs[3] (1 B x 768, nlist 65536, m 96, nprobe 128): with --gpus 8 the sharded run "
                         "(implies --shard-coarse), with --gpus 1 the single-GPU leg of the default line alone")
    ap.add_argument("--flat-rows", type=int, default=10_000_000)
    ap.add_argument("--flat-batch", type=int, default=1024)
    ap.add_argument("--flat-metric", default="l2", choices=["l2", "cosine", "dot"])
    ap.add_argument("--flat-gemm", type=int, default=0, help="mi355_flat_configure gemm_variant (0 = the library's choice)")
    ap.add_argument("--flat-grid", type=int, default=0, help="mi355_flat_configure grid_workgroups")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: keep the exchange on the search stream (MI355_SHARD_NO_OVERLAP) instead of running it "
                         "on the communicator's stream under the next step's scan")
    ap.add_argument("--force-sharded-path", action="store_true",
                    help="dev: run the N > 1 code path (process group, RCCL communicator behind the C ABI, "
                         "mi355_search_sharded, teardown) in a world of one rank: what a 1-GPU box can verify")
    ap.add_argument("--loopback-world", type=int, default=8,
                    help="N = 1 secondary leg: the N-rank sharded search of the same index with all ranks on this one GPU "
                         "(loopback communicator): per-rank stage times of an N-rank step, exchange time, full-size parity "
                         "with the unsharded search (0 / 1 = skip)")
    ap.add_argument("--c5-rows", type=int, default=100_000_000,
                    help="rows of the C5 line (BASELINE.json configs[4]: 100 M x 1536 cosine, refine_factor 10); 0 = skip")
    ap.add_argument("--c4-rows", type=int, default=1_000_000_000,
                    help="rows of the C4 leg (BASELINE.json configs[3]: 1 B x 768, nlist 65536, m 96, nprobe 128) on one GPU; 0 = skip")
    ap.add_argument("--shard-coarse", action="store_true",
                    help="N > 1: shard the coarse quantiser too (MI355_SHARD_COARSE: each rank scores nlist / N centroids; one extra "
                         "gather of nprobe (partition, distance) pairs per query per rank) — the C4 mode")
    ap.add_argument("--batch-per-gpu", type=int, default=0,
                    help="N > 1: queries per step = this x N (the batch grows with the ranks) instead of --batch")
    ap.add_argument("--default-shape-rows", type=int, default=100_000_000,
                    help="rows of the leg at the reference's DEFAULT index shape (rows / 8192 partitions, m = dim / 16, nprobes 20 and 64); 0 = skip")
    ap.add_argument("--widths", type=int, default=1, help="secondary lines at the reference's default PQ widths m = dim / 16 (384-d, 3072-d); 0 = skip")
    ap.add_argument("--gist-rows", type=int, default=1_000_000, help="rows of the GIST1M-shaped recall@1 / latency line; 0 = skip")
    ap.add_argument("--c5-column", default="register", choices=["register", "hostmalloc"],
                    help="C5 host column: the caller's pages registered with hipHostRegister (default) or a hipHostMalloc allocation (A/B of the PCIe gather)")
    ap.add_argument("--c5-hugepages", type=int, default=0, help="back the C5 host column with MADV_HUGEPAGE memory (A/B of the PCIe gather)")
    ap.add_argument("--scan-variant", type=int, default=0)
    ap.add_argument("--slice-rows", type=int, default=0)
    return ap.parse_args()


def main_flat(a):
    """`--workload flat`: the C2 line alone (the default run carries it as `secondary.flat_c2_*`)."""
    if a.gpus != 1:
        raise SystemExit("--workload flat is a single-GPU line (rows shard over ranks through mi355_flat_search_sharded)")
    res = flat_c2(a, a.flat_metric, cpu_queries=32 if a.cpu_seconds > 0 else 0)
    res.update({"n_gpus": 1, "warmup": a.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None})
    import bench_legs as legs
    legs.emit(res)


def main():
    a = parse()
    import bench_legs
    bench_legs.claim_stdout()  # stdout carries the ONE JSON line; everything else this process prints goes to stderr
    if a.workload == "flat":
        return main_flat(a)
    import numpy as np
    import torch
    import torch.distributed as dist

    import bench_legs as legs
    import lancedb_amd
    from lancedb_amd import _abi
    if a.workload == "c4":
        if a.gpus == 1 and not a.force_sharded_path:  # the single-GPU C4 leg of the default line, alone
            torch.cuda.set_device(0)
            res = legs.c4_leg(a, torch, np, torch.device("cuda", 0), n_rows=a.c4_rows, world=a.loopback_world)
            res.update({"n_gpus": 1, "warmup": 2, "higher_is_better": True, "scaling": "strong", "vs_baseline": None})
            legs.emit(res)
            return
        # the sharded run of configs[3]: same code path as the C3 scaling run, C4's shape, sharded coarse stage
        a.n_rows, a.nlist, a.m, a.nprobe, a.shard_coarse = a.c4_rows, 65536, 96, 128, True

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} but WORLD_SIZE={world}: one process per GPU — launch it as\n"
                         f"  python -m torch.distributed.run --nnodes=1 --nproc-per-node {a.gpus} --master-addr 127.0.0.1 --master-port 29533 "
                         f"bench.py --gpus {a.gpus} --steps {a.steps} --warmup {a.warmup}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or a.force_sharded_path
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)  # (torch's eager RCCL init prints the same banner)
        try:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        finally:
            os.dup2(saved, 1)
            os.close(saved)

    n, dim, nlist, m = a.n_rows, a.dim, a.nlist, a.m
    if a.batch_per_gpu and world > 1:
        a.batch = a.batch_per_gpu * world
    # ---- synthetic index: small tables identical on every rank (same seed); the O(rows) arrays are
    # generated PER PARTITION (seed = f(SEED, partition)) for the partitions this rank owns only, so a
    # rank never materialises the others' data and the index is the same for every N (bench_legs.synth_*)
    tables = legs.synth_tables(torch, np, dev, n, dim, nlist, m, a.skew, seed=SEED)
    g, centroids, codebook, lens, part_offsets = (tables[k2] for k2 in ("gen", "centroids", "codebook", "lens", "part_offsets"))
    def plan_owner(n_shards):
        """Partition -> shard.  For more than one shard: balanced by the rows a shard SCANS — the probe histogram of a
        calibration batch drawn from the query distribution (not one of the timed batches) weights the partitions
        (mi355_shard_plan_weighted); every rank takes rank 0's histogram so that all ranks cut the same plan."""
        if n_shards <= 1:
            return lancedb_amd.shard_plan(part_offsets, n_shards)
        gc = torch.Generator(device=dev)
        gc.manual_seed(SEED + 99)
        qc = centroids[torch.randint(0, nlist, (4096,), generator=gc, device=dev)] + 0.5 * torch.randn((4096, dim), generator=gc, device=dev)
        pr = torch.cdist(qc, centroids).topk(a.nprobe, largest=False).indices
        hits = torch.bincount(pr.flatten(), minlength=nlist).to(torch.float32)
        if sharded and world > 1:
            dist.broadcast(hits, src=0)
        return lancedb_amd.shard_plan(part_offsets, n_shards, weights=hits.cpu().numpy())
    owner = plan_owner(world)  # handed to mi355_index_open as part_owner
    legs.synth_rows(torch, np, dev, tables, owner if world > 1 else None, rank)
    codes, row_ids, rows_mine = tables.pop("codes"), tables.pop("row_ids"), tables["rows"]

    t_open = time.time()
    ix = lancedb_amd.IvfPqIndex(centroids, codebook, part_offsets, codes, row_ids, metric="l2",
                                codes_layout=_abi.CODES_PART_TRANSPOSED, device=local_rank,
                                shard_count=world, shard_rank=rank, local_arrays=True, part_owner=owner if world > 1 else None)
    t_open = time.time() - t_open
    rows_local, parts_local = ix.info()
    assert rows_local == rows_mine

    want_cpu = rank == 0 and not sharded and a.cpu_seconds > 0
    h_codes = h_rowids = None
    if want_cpu:
        h_codes = codes.cpu().numpy()
        h_rowids = row_ids.cpu().numpy().astype(np.uint64, copy=False)
    h_centroids = centroids.cpu().numpy()
    h_codebook = codebook.cpu().numpy()
    keep_arrays = rank == 0 and not sharded and a.secondary and a.loopback_world > 1
    if not keep_arrays:  # (the loopback leg cuts its shard handles out of the same device arrays)
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
    # (torch's current stream is normally the null stream, handle 0 = "keep the handle's own stream": the fences below are
    #  device-wide synchronisations, so the timed region is bracketed either way)
    ix.set_stream(stream)
    ix.configure(scan_variant=a.scan_variant, slice_rows=a.slice_rows, profile=0)
    comm = searcher = None
    if sharded:
        # the exchange is RCCL behind the C ABI (mi355_comm_* / mi355_search_sharded: one packed
        # all-gather of the per-shard candidate records on the search stream + a k-way merge on every
        # rank); torch.distributed only carries the 128-byte communicator id to the other ranks
        from lancedb_amd.distributed import Comm, ShardedSearcher, unique_id
        uid = [unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        # (RCCL prints its version banner on stdout when a communicator is created: keep rank 0's stdout to the one JSON line)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            comm = Comm(uid[0], rank, world, device=local_rank)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        searcher = ShardedSearcher(ix, comm, shard_coarse=a.shard_coarse, overlap=not a.no_overlap)

    # SURVEY.md section 8d ends the timed region "on host-visible memory": every step's results (ids, distances, counts: 0.25 MB)
    # are copied to page-locked host memory on the search stream inside the timed region; `value` is that rate.  The same steps
    # with the results left in HBM are timed afterwards and reported as summary.c3_qps_device_io.
    h_out = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in out]

    def step(i, to_host=True):
        r = searcher.search(qpool[i % P], params, out=out) if searcher else ix.search(qpool[i % P], params, out=out)
        if to_host:
            for h, d in zip(h_out, (r.rowids, r.distances, r.counts)):
                h.copy_(d, non_blocking=True)
        return r.rowids, r.distances, r.counts

    def fence():
        torch.cuda.synchronize()
        if sharded:
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
    if sharded:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = ix.stats()

    # roofline of the dominant kernel (ADC scan): algorithmic code bytes per launch
    launches = max(st["scan_launches"], 1)
    bytes_per_launch = st["code_bytes_scanned"] / launches
    us_per_launch = st["us_scan"] / launches
    stat = torch.tensor([st["code_bytes_scanned"], st["us_scan"], float(launches)], dtype=torch.float64, device=dev)
    if sharded:
        dist.all_reduce(stat, op=dist.ReduceOp.SUM)  # whole-job bytes / summed kernel time
    achieved = (float(stat[0]) / float(stat[2])) / (float(stat[1]) / float(stat[2]) * 1e-6) / 1e9 if float(stat[1]) > 0 else 0.0
    qps = a.batch * a.steps / elapsed
    workload = f"ivfpq_{n}x{dim}_nlist{nlist}_m{m}x8_nprobe{a.nprobe}_k{a.k}_l2"
    traffic = traffic_from_profiles(workload, a.batch) if not sharded else None

    result = {
        "metric": ("queries/sec @ recall@10, 100M×768 IVF-PQ nprobe=64 k=10" if a.workload != "c4" else
                   "queries/sec, IVF-PQ 1B×768 nlist=65536 m=96 nprobe=128 k=10, partitions sharded (BASELINE.json configs[3])"),
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
        "roofline": scan_roofline(achieved, bytes_per_launch, us_per_launch, int(float(stat[2])), traffic,
                                  traffic_source_from_profiles(workload, a.batch) if not sharded else None,
                                  torch.cuda.get_device_properties(dev).multi_processor_count,
                                  {s: st["us_" + s] / a.steps for s in ("coarse", "select", "scan", "merge")}),
    }
    if sharded:
        cs = comm.stats()  # identical on every rank: the per-rank scanned rows travel in the gathered slabs
        # every rank's own stage times (HIP events on its search stream) and exchange time, so that a
        # scaling run explains itself: step = max over ranks of (coarse + select + scan + merge) [+ exchange
        # when it is not overlapped with the next step's scan]
        mine_us = {s2: st["us_" + s2] / a.steps for s2 in ("coarse", "select", "scan", "merge")}
        mine_us["exchange_last_step"] = cs["us_exchange"]
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_us)
        result["multi_gpu"] = {
            "exchange": "RCCL behind the C ABI (mi355_search_sharded): one packed all-gather per step + k-way merge",
            "exchange_overlapped_with_next_scan": cs["overlapped"],
            "rccl_ranks": cs["world"], "gathers_per_step": cs["n_gathers"], "bytes_gathered_per_step": cs["bytes_gathered"],
            "rows_scanned_per_rank_timed_steps": cs["rows_scanned"], "load_imbalance_max_over_mean": cs["imbalance"],
            "rows_on_rank": [int(lens[owner == r].sum()) for r in range(world)],
            "shard_plan": "mi355_shard_plan_weighted over the probe histogram of a calibration batch",
            "coarse": "sharded (MI355_SHARD_COARSE)" if a.shard_coarse else "replicated",
            "batch_queries": a.batch, "batch_per_gpu_mode": bool(a.batch_per_gpu),
            "stage_us_per_step_by_rank": per_rank,
        }
        # a scaling run checks itself: the communicator really spans the ranks torch.distributed launched, and every
        # rank returned the same answers (the merge runs on every rank: compare a checksum of the last step's row ids)
        assert cs["world"] == world, f"RCCL communicator spans {cs['world']} ranks, launched {world}"
        chk = torch.stack([last[0].to(torch.int64).sum(), (last[1] * 1e3).to(torch.int64).sum(), last[2].to(torch.int64).sum()])
        lo_, hi_ = chk.clone(), chk.clone()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        result["multi_gpu"]["all_ranks_returned_the_same_results"] = bool((lo_ == hi_).all().item())
        if world == 1:  # --force-sharded-path: the exchange of a world of one must reproduce the plain search
            plain = ix.search(qpool[(a.steps - 1) % P], params)
            torch.cuda.synchronize()
            result["multi_gpu"]["sharded_equals_unsharded"] = bool(
                (plain.rowids == last[0]).all().item() and (plain.distances == last[1]).all().item())

    result["config"]["timed_region"] = "queries resident in HBM -> results in page-locked host memory, every step (SURVEY.md section 8d)"
    if rank == 0 and not sharded:
        # the same K steps with the results left in HBM (what rounds 1-5 quoted as `value`)
        fence()
        t1 = time.perf_counter()
        for i in range(a.steps):
            last = step(i, to_host=False)
        fence()
        dev_io = time.perf_counter() - t1
        result["device_io"] = {"value": a.batch * a.steps / dev_io, "ms_per_step": dev_io / a.steps * 1e3,
                               "timed_region": "queries resident in HBM -> results in HBM"}
    if rank == 0 and not sharded and a.recall_rows > 0:
        result["recall_at_10"] = recall_at_10(a, np, dim, m)
    if rank == 0 and not sharded and a.recall2_rows > 0:
        result["recall_at_10_embedding_like"] = recall_embedding_like(a, np)
    if rank == 0 and not sharded and a.secondary:
        result["secondary"] = {}
        if keep_arrays:
            result["secondary"]["loopback_world%d" % a.loopback_world] = loopback_world(
                a, torch, np, ix, centroids, codebook, part_offsets, codes, row_ids, qpool, params, dev, plan_owner(a.loopback_world))
            del codes, row_ids
            torch.cuda.empty_cache()
        result["secondary"]["latency_c3"] = latency_and_concurrency(a, np, ix, qpool)
        result["secondary"]["concurrent_callers_c3"] = legs.concurrent_callers(
            np, ix, qpool[0].cpu().numpy(), _abi.make_params(k=a.k, nprobe_min=a.nprobe, nprobe_max=a.nprobe), a.k)
        result["secondary"]["qps_vs_batch"] = legs.qps_vs_batch(a, torch, np, ix, centroids, dev)
        ix.set_stream(stream)
        result["secondary"].update(refine_operating_point(a, torch, ix, qpool, rows_local, dim, dev))
        if a.c5_rows > 0:
            result["secondary"]["c5_refine10"] = c5_refine10(a, torch, np, dev)
    if want_cpu:
        result["cpu_baseline"] = cpu_baseline(a, np, h_centroids, h_codebook, part_offsets, h_codes, h_rowids,
                                              qpool[(a.steps - 1) % P], last, params)
    if rank == 0 and not sharded and a.secondary:
        # flat C2 needs 15 GB for its column: drop the 100 M index first
        ix.close()
        del ix
        torch.cuda.empty_cache()
        torch.cuda.empty_cache()
        if a.c4_rows > 0:
            result["secondary"]["c4"] = legs.c4_leg(a, torch, np, dev, n_rows=a.c4_rows, world=a.loopback_world)
        if a.default_shape_rows > 0:
            result["secondary"]["default_shape"] = legs.default_shape_leg(a, torch, np, dev, n_rows=a.default_shape_rows)
        if a.widths:
            result["secondary"].update(legs.width_lines(a, torch, np, dev, n_rows=a.n_rows))
        if a.gist_rows > 0:
            result["secondary"]["gist_like"] = legs.gist_like(a, torch, np, dev, n=a.gist_rows)
        result["secondary"]["c1_flat"] = legs.c1_flat(a, np)
        for metric in ("l2", "cosine"):
            result["secondary"]["flat_c2_" + metric] = flat_c2(a, metric, cpu_queries=32 if metric == "l2" else 16)
    if rank == 0 and "recall_at_10" in result:
        # every row names ONE index: QPS and recall@10 measured on the same trained index (recall_at_10.points); the headline
        # line above stays the 100 M synthetic index (uniform random codes: its recall is meaningless)
        result["qps_vs_recall_at_10"] = result["recall_at_10"].get("points", [])
    if rank == 0:
        result["summary"] = legs.summary_of(result)
        legs.emit(result)  # full document -> bench_detail.json; stdout gets ONE line of < 4 KB (contract keys + roofline + cpu_baseline + summary)
    if sharded:
        # tear down in dependency order: the communicator before the index whose stream it used
        torch.cuda.synchronize()
        dist.barrier()
        comm.close()
        ix.close()
        dist.destroy_process_group()


def scan_roofline(achieved, bytes_per_launch, us_per_launch, launches, traffic, traffic_source, n_cus, stage_us):
    """`frac` is the NO-REUSE algorithmic byte rate (m bytes per scanned row per query)
    over the scan kernel's own time against HBM peak — it exceeds 1 because the XCD-local work queues
    serve a partition's codes to many queries from L2.  The ceilings that actually bind the kernel
    are reported beside it: the LDS gather rate (one 4-byte table gather per code byte; 32 gathers
    per clock per CU conflict-free, MI355X_MICROARCH.md §LDS) and the measured HBM traffic."""
    gathers_per_s = bytes_per_launch / (us_per_launch * 1e-6) if us_per_launch else 0.0
    gather_peak = n_cus * 32 * 2.4e9
    r = {
        "bound": "hbm", "kernel": "k_scan (LUT build + ADC scan + top-k)",
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "frac_definition": "no-reuse algorithmic code bytes / scan-kernel time / 8 TB/s (SURVEY.md §8d); > 1 = cross-query reuse in L2",
        "traffic": traffic, "traffic_source": traffic_source,
        "hbm_measured_frac": (traffic / (us_per_launch * 1e-6) / 1e9 / HBM_PEAK_GBS) if (traffic and us_per_launch) else None,
        "lds_gather": {"achieved": gathers_per_s / 1e12, "peak": gather_peak / 1e12, "unit": "T gathers/s",
                       "frac": gathers_per_s / gather_peak,
                       "peak_definition": f"{n_cus} CUs x 32 ds_read_b32 lanes/clk x 2.4 GHz, conflict-free"},
        "algorithmic_bytes_per_launch": bytes_per_launch, "us_per_launch": us_per_launch,
        "launches": launches, "stage_us_per_step": stage_us,
    }
    return r


def latency_and_concurrency(a, np, ix, qpool):
    """Host-I/O single-query latency on the 100 M index from one thread (callers are tokio workers,
    python/src/runtime.rs:31-37), eager launches and a replayed hipGraph (MI355_CFG_GRAPH); per-stage device times of
    one query.  Concurrent callers: bench_legs.concurrent_callers."""
    from lancedb_amd import _abi
    hq = qpool[0].cpu().numpy()
    kw = dict(k=a.k, nprobe_min=a.nprobe, nprobe_max=a.nprobe)
    out = {}
    for mode, graph in (("graph", True), ("eager", False)):
        ix.configure(profile=0, graph=graph, coalesce=False)
        for i in range(8):
            ix.search(hq[i:i + 1], **kw)
        lat = []
        for i in range(200):
            t0 = time.perf_counter()
            ix.search(hq[i % 512:i % 512 + 1], **kw)
            lat.append(time.perf_counter() - t0)
        lat = np.sort(np.array(lat)) * 1e6
        out[f"single_query_us_{mode}"] = {"p50": float(lat[100]), "p99": float(lat[197]), "mean": float(lat.mean())}
        if graph:
            out["graph_replays"] = ix.stats()["graph_replays"]  # (configure() resets the counters)
    # where a single query's time goes on the device (per-stage HIP events of one eager search)
    ix.configure(profile=1, graph=False, coalesce=False)
    ix.search(hq[7:8], **kw)
    st = ix.stats()
    out["single_query_stage_us"] = {s2: st["us_" + s2] for s2 in ("coarse", "select", "scan", "merge")}
    out["single_query_stage_us_note"] = ("intervals between HIP events recorded around the four launches: every marker adds ~5 us of queue time, so the "
                                         "stages sum to MORE than the eager p50 beside them; the kernel trace of the same searches (rocprofv3 --kernel-trace, "
                                         "committed: profiles/r06_o_kernel_trace_latency_paths_final_tree.txt) is 10.1 + 16.5 + 59.7 + 9.0 = 95.5 us back to back")
    # (concurrent callers: secondary.concurrent_callers_c3 — C++ threads; Python threads measured the interpreter lock)
    ix.configure(profile=0, graph=False, coalesce=True)  # the defaults of a freshly opened handle
    return out


def refine_operating_point(a, torch, ix, qpool, n_rows, dim, dev):
    """The SAME 100 M index with refine_factor 10 and 25 (query.rs:1302-1332): k * rf ANN candidates per query,
    exact re-rank on raw bf16 vectors resident in HBM (100 M x 768 x 2 B = 154 GB, borrowed from the
    caller: mi355_index_attach_raw).  The recall of each point on a trained index is in the recall
    leg (recall_at_10.nprobe64_refine10 / _refine25); the raw vectors here are random finite bf16
    values (the gather / exact-distance work does not depend on them)."""
    import lancedb_amd
    from lancedb_amd import _abi
    free, _ = torch.cuda.mem_get_info(dev)
    need = n_rows * dim * 2
    if free < need + (24 << 30):
        return {"c3_refine10": {"skipped": f"{need / 1e9:.0f} GB of raw vectors do not fit ({free / 1e9:.0f} GB free)"}}
    raw = torch.empty((n_rows, dim), dtype=torch.int16, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(SEED + 7)
    step = 8_000_000
    for r0 in range(0, n_rows, step):  # bf16 bit patterns below 0x4000: finite values in (0, 2)
        raw[r0:r0 + step].random_(0, 0x4000, generator=g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ix.attach_raw_vectors(raw, _abi.DTYPE_BF16)
    t_attach = time.perf_counter() - t0
    B, k = a.batch, a.k
    out = (torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
           torch.empty((B,), dtype=torch.int32, device=dev))
    res = {}
    for rf in (10, 25):  # k * rf = 100 / 250 candidates per query (recall of both: recall_at_10.nprobe64_refine*)
        params = _abi.make_params(k=k, nprobe_min=a.nprobe, nprobe_max=a.nprobe, refine_factor=rf)
        ix.configure(profile=0)
        for i in range(2):
            ix.search(qpool[i % len(qpool)], params, out=out)
        torch.cuda.synchronize()
        ix.configure(profile=2)
        steps = max(3, a.steps // 2)
        t0 = time.perf_counter()
        for i in range(steps):
            ix.search(qpool[i % len(qpool)], params, out=out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = ix.stats()
        refine_bytes = B * k * rf * dim * 2  # algorithmic: k * rf raw rows per query
        res[f"c3_refine{rf}"] = {
            "value": B * steps / dt, "unit": "queries/s", "ms_per_step": dt / steps * 1e3, "steps": steps,
            "config": {"workload": f"ivfpq C3 + refine_factor {rf}, raw bf16 vectors resident", "k": k, "refine_factor": rf,
                       "nprobe": a.nprobe, "batch_queries": B, "raw_vectors_gb": need / 1e9, "attach_s": round(t_attach, 2)},
            "stage_us_per_step": {s2: st["us_" + s2] / steps for s2 in ("coarse", "select", "scan", "merge", "refine")},
            "refine_gather": {"algorithmic_bytes_per_step": refine_bytes,
                              "gb_per_s": refine_bytes / max(st["us_refine"] / steps, 1e-9) / 1e3},
            "recall_at_10": f"see recall_at_10.nprobe64_refine{rf} (trained index)"}
    ix.configure(profile=0)
    ix.detach_raw_vectors()
    del raw
    torch.cuda.empty_cache()
    return res



def loopback_world(a, torch, np, ix, centroids, codebook, part_offsets, codes, row_ids, qpool, params, dev, owner):
    """The N-rank sharded search of the SAME index with every rank on this GPU (mi355_comm_create_loopback:
    one thread and one shard handle per rank, the gather = device copies into the slab layout ncclAllGather
    fills).  Three things a 1-GPU box can measure about an N-GPU step: (1) each rank's own stage times with
    the GPU to itself (its partitions' scan, the replicated coarse / select / plan, its local merge);
    (2) the exchange (gather + merge of world slabs) on the communicator's stream; (3) that all N ranks
    together return the unsharded result at full size.  `step_model_ms` = the slowest rank's stages (the
    exchange overlaps the next step's scan) — a model of the N-GPU step built from measured terms, not a
    measurement of N GPUs."""
    import lancedb_amd
    from lancedb_amd import _abi
    from lancedb_amd.distributed import Comm, ShardedSearcher, run_ranks
    world, B, k, P = a.loopback_world, a.batch, a.k, len(qpool)
    t0 = time.perf_counter()
    shards = [lancedb_amd.IvfPqIndex(centroids, codebook, part_offsets, codes, row_ids, metric="l2",
                                     codes_layout=_abi.CODES_PART_TRANSPOSED, shard_count=world, shard_rank=r, part_owner=owner)
              for r in range(world)]
    t_open = time.perf_counter() - t0
    comms = Comm.loopback(world)
    outs = [(torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
             torch.empty((B,), dtype=torch.int32, device=dev)) for _ in range(world)]
    steps = max(4, a.steps // 2)
    torch.cuda.synchronize()
    # (1) one rank at a time: its stages with the GPU to itself (after one untimed round over all ranks: the first
    # handle measured otherwise also pays for cold caches and clocks)
    for r in range(world):
        for i in range(2):
            shards[r].search(qpool[i % P], params, out=outs[r])
        shards[r].sync()
    per_rank = []
    for r in range(world):
        # (the plain search of a shard handle = its local stages: replicated coarse / select / plan, the scan of
        # the probed partitions it owns, its local merge)
        shards[r].configure(profile=0)
        shards[r].search(qpool[0], params, out=outs[r])
        shards[r].sync()
        shards[r].configure(profile=2)
        t1 = time.perf_counter()
        for i in range(steps):
            shards[r].search(qpool[i % P], params, out=outs[r])
        shards[r].sync()
        dt = (time.perf_counter() - t1) / steps
        st = shards[r].stats()
        per_rank.append({"rows": shards[r].info()[0], "ms_per_step_wall": dt * 1e3,
                         **{s2 + "_us": st["us_" + s2] / steps for s2 in ("coarse", "select", "scan", "merge")}})
        shards[r].configure(profile=0)
    # (2) + (3) all ranks together, one thread each
    ref = ix.search(qpool[(steps - 1) % P], params)
    torch.cuda.synchronize()
    modes = {}
    for mode, overlap in (("overlapped", True), ("serial", False)):
        def rank_fn(r):
            sh = ShardedSearcher(shards[r], comms[r], overlap=overlap)
            sh.search(qpool[0], params, out=outs[r])
            shards[r].sync()
            t2 = time.perf_counter()
            for i in range(steps):
                sh.search(qpool[i % P], params, out=outs[r])
            shards[r].sync()
            return (time.perf_counter() - t2) / steps, comms[r].stats()
        got = run_ranks([lambda r=r: rank_fn(r) for r in range(world)])
        same = all(bool((outs[r][0] == ref.rowids).all().item() and (outs[r][1] == ref.distances).all().item())
                   for r in range(world))
        cs = got[0][1]
        modes[mode] = {"all_ranks_on_one_gpu_ms_per_step": max(g[0] for g in got) * 1e3,
                       "exchange_us_by_rank": [g[1]["us_exchange"] for g in got], "gathers_per_step": cs["n_gathers"],
                       "bytes_gathered_per_step": cs["bytes_gathered"], "rows_scanned_by_rank": cs["rows_scanned"],
                       "load_imbalance_max_over_mean": cs["imbalance"], "every_rank_equals_unsharded": same}
    stages = [p["ms_per_step_wall"] * 1e3 for p in per_rank]  # wall per step of one rank alone: its stages + planner + launch gaps
    res = {"world": world, "shard_open_s": round(t_open, 2), "steps": steps, "batch_queries": B,
           "shard_plan": "mi355_shard_plan_weighted over the probe histogram of a calibration batch (rows held per rank: "
                         + ", ".join(str(p["rows"]) for p in per_rank) + ")",
           "stage_us_per_step_by_rank_alone": per_rank, **modes,
           "step_model": {"slowest_rank_us": max(stages), "mean_rank_us": float(np.mean(stages)),
                          "overlapped_ms": max(stages) / 1e3, "qps_overlapped": B / (max(stages) * 1e-6),
                          "qps_if_ranks_were_balanced": B / (float(np.mean(stages)) * 1e-6),
                          "note": "per-rank wall time of a step measured with the GPU to one rank; with the exchange on the "
                                  "communicator's stream the N-GPU step is the slowest rank's own stages.  exchange_us_by_rank above is "
                                  "the span first gather .. final merge with N ranks time-sharing ONE GPU — it mostly waits for "
                                  "the other ranks' scans; its own cost is the world-of-one RCCL figure in "
                                  "profiles/r03_a_bench_sharded_world1_*.json (58 us) plus the copies of N slabs.  A model from "
                                  "measured terms, not a measurement of N GPUs; divide qps by this run's N = 1 value for the "
                                  "modelled scaling efficiency"}}
    for c in comms:
        c.close()
    for s2 in shards:
        s2.close()
    torch.cuda.empty_cache()
    return res


def host_column(np, n, dim, hugepages):
    """The caller's raw column of the C5 leg ([n, dim] bf16 bit patterns in host memory).  hugepages: anonymous memory
    advised MADV_HUGEPAGE before it is touched — the GPU reaches a registered host range through the IOMMU / its own
    page tables, and a random 3-KiB row per candidate is a translation miss per row with 4-KiB pages."""
    if not hugepages:
        return np.empty((n, dim), dtype=np.uint16)
    import mmap
    nbytes = n * dim * 2
    mm = mmap.mmap(-1, (nbytes + (2 << 20) - 1) & ~((2 << 20) - 1), flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    try:
        mm.madvise(mmap.MADV_HUGEPAGE)
    except (AttributeError, OSError):
        pass
    return np.frombuffer(mm, dtype=np.uint16, count=n * dim).reshape(n, dim)


def c5_refine10(a, torch, np, dev):
    """BASELINE.json configs[4] (query.rs:1302-1332, table/query.rs:311-313): IVF-PQ + refine, 100 M x 1536, cosine,
    nprobe 64, refine_factor 10; nlist 4096 and m = 96 = dim / 16 are the reference's default rules
    (index/vector.rs:306-310) — BASELINE names neither.  The PQ codes live in HBM; the raw bf16 column is 307 GB,
    more than the 288 GB of HBM, so it stays in HOST memory, page-locked and mapped (HostMappedArray ->
    mi355_index_attach_raw): the refine kernel gathers k * refine_factor = 100 rows of 3 KiB per query over PCIe.
    The column takes at most 70 % of the host memory the process may still use (MemAvailable and the container's
    cgroup limit); when that is less than 100 M rows the line runs on the largest row count that fits and names it in
    `config.workload` / `rows_note`; with no usable host memory it falls back to an HBM-resident column."""
    import threading

    import lancedb_amd
    from lancedb_amd import _abi
    n, dim, nlist, m, nprobe, k, rf, B = a.c5_rows, 1536, 4096, 96, a.nprobe, a.k, 10, a.batch
    row_bytes = dim * 2
    # host memory this PROCESS may still take: MemAvailable, and the container's cgroup limit when there is one
    # (the MI355X boxes of this pool: 3 TB of RAM behind a 300 GiB cgroup — a 307 GB column gets the box killed)
    avail = 0
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable:"):
            avail = int(line.split()[1]) * 1024
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        cur = int(open("/sys/fs/cgroup/memory.current").read().strip())
        if lim != "max":
            avail = min(avail, int(lim) - cur)
    except (OSError, ValueError):
        try:
            avail = min(avail, int(open("/sys/fs/cgroup/memory/memory.limit_in_bytes").read()) -
                        int(open("/sys/fs/cgroup/memory/memory.usage_in_bytes").read()))
        except (OSError, ValueError):
            pass
    n_full = n
    budget = int(avail * 0.7) - (8 << 30)  # the column may take 70 % of what is left (+ the oracle leg's copy of the codes)
    host_mapped = budget >= 20_000_000 * row_bytes
    if host_mapped:
        n = min(n, budget // (row_bytes + m + 8))
    else:
        free_hbm, _ = torch.cuda.mem_get_info(dev)
        n_fit = int((free_hbm - (40 << 30)) // (row_bytes + 2 * m + 16))
        if n_fit < 1_000_000:
            return {"skipped": f"neither host RAM ({avail / 1e9:.0f} GB usable) nor HBM for a raw column"}
        n = min(n, n_fit)
    n = int(n) // 1_000_000 * 1_000_000
    t_build = time.perf_counter()
    g = torch.Generator(device=dev)
    g.manual_seed(SEED + 5)
    centroids = torch.randn((nlist, dim), generator=g, device=dev, dtype=torch.float32)
    centroids /= centroids.norm(dim=1, keepdim=True)  # cosine: the coarse quantiser sees unit vectors
    codebook = torch.randn((m, 256, dim // m), generator=g, device=dev, dtype=torch.float32) * (0.5 / np.sqrt(dim))
    rng = np.random.default_rng(SEED + 5)
    w = np.exp(rng.normal(0.0, a.skew, size=nlist))
    lens = rng.multinomial(n, w / w.sum())
    part_offsets = np.zeros(nlist + 1, dtype=np.uint64)
    part_offsets[1:] = np.cumsum(lens)
    codes = torch.randint(0, 256, (n * m,), generator=g, device=dev, dtype=torch.uint8)  # lance's transposed blocks
    mult = 982_451_653
    while np.gcd(mult, n) != 1:
        mult += 2
    row_ids = (torch.arange(n, device=dev, dtype=torch.int64) * mult + 12_345) % n
    torch.cuda.synchronize()
    ix = lancedb_amd.IvfPqIndex(centroids, codebook, part_offsets, codes, row_ids, metric="cosine",
                                codes_layout=_abi.CODES_PART_TRANSPOSED)
    # raw vectors: bf16 bit patterns of finite values in (0, 2) (the gather and the exact-distance work do not
    # depend on the values); host: filled by 64 threads from one random block
    t_raw = time.perf_counter()
    if host_mapped:
        col_alloc = None
        if getattr(a, "c5_column", "register") == "hostmalloc":  # dev A/B: pages allocated AND mapped by the HIP runtime
            col_alloc = lancedb_amd.HostAllocArray((n, dim), np.uint16)
            raw = col_alloc.host
        else:
            raw = host_column(np, n, dim, a.c5_hugepages)
        flat = raw.reshape(-1)
        blk = np.random.default_rng(SEED + 6).integers(0, 0x4000, size=1 << 27, dtype=np.uint16)  # 256 MB

        def fill(lo, hi):
            for o in range(lo, hi, blk.size):
                e = min(hi, o + blk.size)
                flat[o:e] = blk[:e - o]
        T = 64
        th = [threading.Thread(target=fill, args=(flat.size * i // T, flat.size * (i + 1) // T)) for i in range(T)]
        [t.start() for t in th]
        [t.join() for t in th]
        t_fill = time.perf_counter() - t_raw
        t_raw = time.perf_counter()
        col = col_alloc if col_alloc is not None else lancedb_amd.HostMappedArray(raw)
        t_map = time.perf_counter() - t_raw
    else:
        col = torch.empty((n, dim), dtype=torch.int16, device=dev)
        gg = torch.Generator(device=dev)
        gg.manual_seed(SEED + 6)
        for r0 in range(0, n, 8_000_000):
            col[r0:r0 + 8_000_000].random_(0, 0x4000, generator=gg)
        torch.cuda.synchronize()
        t_fill, t_map = time.perf_counter() - t_raw, 0.0
    ix.attach_raw_vectors(col, _abi.DTYPE_BF16)
    t_build = time.perf_counter() - t_build
    P = 3
    qpool = []
    for _ in range(P):
        pick = torch.randint(0, nlist, (B,), generator=g, device=dev)
        qpool.append((centroids[pick] + (0.5 / np.sqrt(dim)) * torch.randn((B, dim), generator=g, device=dev)).contiguous())
    params = _abi.make_params(k=k, nprobe_min=nprobe, nprobe_max=nprobe, refine_factor=rf)
    out = (torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
           torch.empty((B,), dtype=torch.int32, device=dev))
    stream = torch.cuda.current_stream().cuda_stream
    ix.set_stream(stream)
    # the re-rank of step i (a PCIe gather) runs beside the scan of step i + 1: opt-in (MI355_CFG_DEFER_REFINE), results
    # complete at sync()
    ix.configure(profile=0, defer_refine=host_mapped)
    for i in range(2):
        ix.search(qpool[i % P], params, out=out)
    ix.sync()
    torch.cuda.synchronize()
    ix.configure(profile=2)
    steps = max(6, a.steps)  # (the re-rank of step i runs beside the scan of step i + 1: the last one drains inside the timed region)
    t0 = time.perf_counter()
    for i in range(steps):
        last = ix.search(qpool[i % P], params, out=out)
    ix.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = ix.stats()
    refine_bytes = B * k * rf * row_bytes
    res = {
        "metric": "queries/sec, IVF-PQ + refine 100M×1536 cosine nprobe=64 refine_factor=10 (BASELINE.json configs[4])",
        "value": B * steps / dt, "unit": "queries/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"ivfpq_{n}x{dim}_nlist{nlist}_m{m}x8_nprobe{nprobe}_k{k}_cosine_refine{rf}_raw_bf16_"
                               + ("host_mapped" if host_mapped else "hbm_resident") + ("" if n == n_full else "_reduced_rows"),
                   "rows_asked": n_full,
                   "rows_note": None if n == n_full else
                   f"{n_full} x {dim} bf16 = {n_full * row_bytes / 1e9:.0f} GB does not fit what this box lets the process use "
                   f"({avail / 1e9:.0f} GB of host memory: MemAvailable / cgroup limit); the column is the largest that takes 70 % of it",
                   "n_rows": n, "dim": dim, "nlist": nlist, "m": m, "nprobe": nprobe, "k": k, "refine_factor": rf,
                   "batch_queries": B, "raw_vectors_gb": n * row_bytes / 1e9,
                   "raw_vectors": "host memory, page-locked + mapped, gathered over PCIe" if host_mapped else "HBM (host RAM too small)",
                   "host_mem_usable_gb": round(avail / 1e9), "build_s": round(t_build, 1), "raw_fill_s": round(t_fill, 1),
                   "raw_page_lock_s": round(t_map, 2)},
        "stage_us_per_step": {s2: st["us_" + s2] / steps for s2 in ("coarse", "select", "plan", "scan", "merge", "refine")},
        "lut_images": st["lut_images"],
        "refine_gather": {"algorithmic_bytes_per_step": refine_bytes, "gb_per_s": refine_bytes / max(st["us_refine"] / steps, 1e-9) / 1e3,
                          "peak": "PCIe Gen5 x16 ≈ 64 GB/s per direction" if host_mapped else "HBM 8 TB/s",
                          "rows_per_step": B * k * rf, "row_bytes": row_bytes},
        "scan_roofline": {"algorithmic_gb_per_s": st["code_bytes_scanned"] / max(st["us_scan"], 1e-9) / 1e3,
                          "frac_of_8tbs": st["code_bytes_scanned"] / max(st["us_scan"], 1e-9) / 1e3 / HBM_PEAK_GBS}}
    if a.cpu_seconds > 0:
        from oracle import oracle as orc
        orc.build()
        import bench_legs as legs
        hc = legs.host_cores()
        cores = hc["usable"]
        nq = min(B, 256)  # (the parity sample does not shrink with the container's CPU quota)
        h_raw = raw if host_mapped else col.cpu().numpy().view(np.uint16)
        ox = orc.OracleIndex(centroids.cpu().numpy(), codebook.cpu().numpy(), part_offsets, codes.cpu().numpy(),
                             row_ids.cpu().numpy().astype(np.uint64), raw_vectors=h_raw, raw_dtype=_abi.DTYPE_BF16,
                             metric="cosine", codes_layout=1, borrow=True)
        hq = qpool[(steps - 1) % P][:nq].cpu().numpy()
        t1 = time.perf_counter()
        ids, dist, cnt, _ = ox.search(hq, params, nthreads=cores)
        t_cpu = time.perf_counter() - t1
        g_ids = last.rowids[:nq].cpu().numpy().astype(np.uint64)
        g_dist = last.distances[:nq].cpu().numpy()
        res["cpu_baseline"] = {"value": nq / t_cpu, "unit": "queries/s", "cores": cores, "host_cores": hc, "kind": "port",
                               "stage_cpu_seconds": ox.last_stage_seconds,
                               "sample": f"{nq} queries of the last timed batch, one per thread, {t_cpu:.1f} s; C restatement "
                                         "(oracle/ann_oracle.c), not the reference binary",
                               "parity": {"queries": nq, "rowids_bit_exact": bool((g_ids == ids).all()),
                                          "max_rel_distance_error": float(np.max(np.abs(g_dist - dist) / np.maximum(np.abs(dist), 1e-30)))}}
        ox.close()
    ix.detach_raw_vectors()
    ix.close()
    if host_mapped:
        col.close()
        del raw
    del col, codes, row_ids
    torch.cuda.empty_cache()
    return res


def flat_c2(a, metric, cpu_queries):
    """BASELINE.json configs[1]: flat KNN over 10 M x 768 bf16, 1024 queries per step, as the bf16 MFMA
    GEMM filter + exact re-rank.  `roofline` is the GEMM kernel's own (HIP events around its launches,
    mi355_flat_last_stats); the CPU leg sweeps the whole column for `cpu_queries` queries of the last
    timed batch and doubles as the full-size parity check."""
    import numpy as np
    import torch

    import lancedb_amd
    from lancedb_amd import _abi
    dev = torch.device("cuda", 0)
    n, dim, B, k = a.flat_rows, a.dim, a.flat_batch, a.k
    g = torch.Generator(device=dev)
    g.manual_seed(SEED)
    col = torch.empty((n, dim), device=dev, dtype=torch.bfloat16)
    step = 2_000_000
    for r0 in range(0, n, step):
        col[r0:r0 + step] = torch.randn((min(step, n - r0), dim), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    P = 2
    qpool = [torch.randn((B, dim), generator=g, device=dev, dtype=torch.float32) for _ in range(P)]
    torch.cuda.synchronize()
    fl = lancedb_amd.FlatIndex(col.view(torch.int16), dtype=_abi.DTYPE_BF16, device=0)
    stream = torch.cuda.current_stream().cuda_stream
    fl.set_stream(stream)
    mt = _abi.METRIC_NAMES[metric]
    params = _abi.make_params(k=k, nprobe_min=1, nprobe_max=1, metric=mt)
    out = (torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
           torch.empty((B,), dtype=torch.int32, device=dev))
    # (path pinned: this line measures the GEMM filter; at the full C2 size the library's own per-call choice is the same)
    fl.configure(gemm_variant=a.flat_gemm, grid_workgroups=a.flat_grid, path="filter")
    for i in range(max(a.warmup, 1)):
        fl.search(qpool[i % P], params, out=out)
    torch.cuda.synchronize()
    fl.configure(gemm_variant=a.flat_gemm, grid_workgroups=a.flat_grid, profile=True, path="filter")
    steps = a.steps
    t0 = time.perf_counter()
    for i in range(steps):
        last = fl.search(qpool[i % P], params, out=out)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert fl.info() == (1, 1), "the MFMA filter path did not run"
    fs = fl.stats()
    flops = 2.0 * B * n * dim
    gemm_tf = fs["gemm_flops"] / max(fs["us_gemm"], 1e-9) / 1e6
    res = {
        "metric": "queries/sec, flat (no index) KNN 10M×768 bf16, batch 1024, k=10 (BASELINE.json configs[1])",
        "value": B * steps / elapsed, "unit": "queries/s", "steps": steps, "ms_per_step": elapsed / steps * 1e3,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"flat_{n}x{dim}_bf16_batch{B}_k{k}_{metric}", "n_rows": n, "dim": dim, "batch_queries": B, "k": k,
                   "metric": metric, "gemm_variant": fs["gemm_variant"]},
        "roofline": {"bound": "mfma", "kernel": "k_flat_gemm (bf16 MFMA filter GEMM, own HIP events)",
                     "achieved": gemm_tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": gemm_tf / 2500.0,
                     "us_per_launch": fs["us_gemm"] / max(fs["gemm_launches"], 1), "launches": fs["gemm_launches"],
                     "algorithmic_flops_per_launch": fs["gemm_flops"] / max(fs["gemm_launches"], 1),
                     "rest_of_step_us": fs["us_rest"] / max(fs["gemm_launches"], 1),
                     "whole_step_tflops": flops / (elapsed / steps) / 1e12,
                     "traffic": traffic_from_profiles(f"flat_{n}x{dim}_bf16_batch{B}_k{k}_l2", B),
                     "traffic_source": traffic_source_from_profiles(f"flat_{n}x{dim}_bf16_batch{B}_k{k}_l2", B)},
    }
    if a.cpu_seconds > 0 and cpu_queries:
        from oracle import oracle as orc
        orc.build()
        import bench_legs as legs
        hc = legs.host_cores()
        cores = hc["usable"]
        nq = min(B, cpu_queries)  # every query sweeps the whole column on the host
        hv = col.view(torch.int16).cpu().numpy().view(np.uint16)
        hq = qpool[(steps - 1) % P][:nq].cpu().numpy()
        t1 = time.perf_counter()
        ids, dist, cnt, _ = orc.flat_search(hv, hq, k=k, dtype=_abi.DTYPE_BF16, metric=mt)
        dt = time.perf_counter() - t1
        g_ids = last.rowids[:nq].cpu().numpy().astype(np.uint64)
        g_dist = last.distances[:nq].cpu().numpy()
        res["cpu_baseline"] = {
            "value": nq / dt, "unit": "queries/s", "cores": min(cores, nq), "host_cores": hc, "kind": "port",
            "sample": f"{nq} queries of the last timed batch, one per thread, {dt:.1f} s on {min(cores, nq)} of {cores} usable host cores; "
                      "C restatement (oracle/ann_oracle.c), not the reference binary",
            "parity": {"queries": nq, "rowids_bit_exact": bool((g_ids == ids).all()),
                       "distances_equal": bool((g_dist == dist).all())}}
    fl.close()
    del fl, col
    torch.cuda.empty_cache()
    return res


def recall_nlist(a):
    """IVF partitions of the trained index: the C3 value (4096) from 8 M rows on, 1024 below (toy runs)."""
    return 4096 if a.recall_rows >= 8_000_000 else 1024


def recall_column(a, dim, data):
    """The raw column and held-out queries of the recall legs, on the device.
      "embedding": unit vectors of intrinsic dimension 48 with a power-law spectrum, x = normalise(z W + 2 % noise), z from a
                   2000-cluster mixture — the shape of text-embedding columns; neighbours are separated by more than the PQ
                   error, so nprobes AND refine_factor move recall (round 6: the one-index leg uses this set);
      "mixture":   the isotropic 4096-component Gaussian mixture of rounds 1-5 (true top-10 within the PQ error of each other:
                   recall without refine is PQ-limited, 0.21 at any nprobes) — kept for tests/tools/parity_exposure*.py."""
    import torch
    n, nq = a.recall_rows, a.recall_queries
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    if data == "mixture":
        g.manual_seed(SEED)
        n_comp = 4096
        comps = torch.randn((n_comp, dim), generator=g, device=dev) * 1.5
        x = torch.empty((n, dim), device=dev)
        for r0 in range(0, n, 500_000):
            c = min(500_000, n - r0)
            x[r0:r0 + c] = comps[torch.randint(0, n_comp, (c,), generator=g, device=dev)] + torch.randn((c, dim), generator=g, device=dev)
        q = comps[torch.randint(0, n_comp, (nq,), generator=g, device=dev)] + torch.randn((nq, dim), generator=g, device=dev)
        return x, q, g, f"{n_comp}-component isotropic Gaussian mixture"
    import numpy as np
    g.manual_seed(SEED + 21)
    idim, n_cl = 48, 2000
    spec = 1.0 / torch.sqrt(1.0 + torch.arange(idim, device=dev, dtype=torch.float32))
    W = torch.linalg.qr(torch.randn((dim, idim), generator=g, device=dev))[0].T.contiguous()  # [idim, dim], orthonormal rows
    centers = torch.randn((n_cl, idim), generator=g, device=dev)

    def draw(cnt):
        z = (centers[torch.randint(0, n_cl, (cnt,), generator=g, device=dev)] + 0.35 * torch.randn((cnt, idim), generator=g, device=dev)) * spec
        v = z @ W + 0.02 * torch.randn((cnt, dim), generator=g, device=dev) * float(spec.norm()) / np.sqrt(dim)
        return v / v.norm(dim=1, keepdim=True)
    x = torch.empty((n, dim), device=dev)
    for r0 in range(0, n, 500_000):
        x[r0:r0 + 500_000] = draw(min(500_000, n - r0))
    return x, draw(nq), g, f"embedding-like: unit vectors of intrinsic dimension {idim} (power-law spectrum) from a {n_cl}-cluster mixture + 2 % isotropic noise"


def train_and_encode(a, x, g, nlist, m):
    """IVF centroids (sample_rate 256 rows per partition) and residual PQ codebooks (256 x 256 rows) trained for `--recall-iters`
    Lloyd iterations, all rows encoded — by the engine's own build entry points (mi355_kmeans_train / mi355_ivf_residuals /
    mi355_pq_train / mi355_ivfpq_encode; parameters as the reference's builder, rust/lancedb/src/index/vector.rs:61-119, :266-319)."""
    import torch
    import lancedb_amd
    n, dim = x.shape
    dsub, iters, dev = dim // m, a.recall_iters, x.device
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
    cb0 = seeds.reshape(256, m, dsub).permute(1, 0, 2).contiguous()
    torch.cuda.synchronize()
    codebook = lancedb_amd.pq_train(resid, cb0, iters=iters)
    torch.cuda.synchronize()
    t_train = time.perf_counter() - t_train
    del ivf_rows, pq_rows, resid
    t_enc = time.perf_counter()
    part_offsets, codes, order = lancedb_amd.ivfpq_encode(x, cen, codebook)
    torch.cuda.synchronize()
    t_enc = time.perf_counter() - t_enc
    # the raw column the re-rank reads: bf16 in index order, resident in HBM (what a 100 M x 768 column has to be: 154 GB)
    xs = torch.empty((n, dim), dtype=torch.bfloat16, device=dev)
    for r0 in range(0, n, 1_000_000):
        xs[r0:r0 + 1_000_000] = x[order[r0:r0 + 1_000_000].to(torch.int64)].to(torch.bfloat16)
    torch.cuda.synchronize()
    return {"cen": cen, "codebook": codebook, "part_offsets": part_offsets, "codes": codes, "order": order, "xs": xs,
            "t_train": t_train, "t_enc": t_enc, "nlist": nlist, "m": m}


def recall_index(a, dim, m, data="mixture"):
    """One trained index over a recall column (tests/tools/parity_exposure*.py, trained_dev_counters.py); everything stays on the device."""
    x, q, g, desc = recall_column(a, dim, data)
    R = train_and_encode(a, x, g, recall_nlist(a), m)
    R.update({"x": x, "q": q, "n_comp": 4096, "data": desc})
    return R


def host_truth_top10(np, x, hq, chunk=1_000_000):
    """Exact top-10 (L2) of the host queries `hq` over the device column `x`, computed ON THE HOST with numpy's sgemm, chunk by
    chunk — a truth that shares no code with the engine (the recall "truth" was the engine's own flat path)."""
    nq = hq.shape[0]
    best_d = np.full((nq, 10), np.inf, np.float32)
    best_i = np.zeros((nq, 10), np.int64)
    qq = (hq.astype(np.float64) ** 2).sum(1).astype(np.float32)
    for r0 in range(0, x.shape[0], chunk):
        xc = x[r0:r0 + chunk].cpu().numpy()
        d = (xc * xc).sum(1)[None, :] - 2.0 * (hq @ xc.T) + qq[:, None]
        k = min(10, d.shape[1])
        part = np.argpartition(d, k - 1, axis=1)[:, :k]
        pd = np.take_along_axis(d, part, axis=1)
        cat_d = np.concatenate([best_d, pd], axis=1)
        cat_i = np.concatenate([best_i, part + r0], axis=1)
        sel = np.argsort(cat_d, axis=1, kind="stable")[:, :10]
        best_d, best_i = np.take_along_axis(cat_d, sel, axis=1), np.take_along_axis(cat_i, sel, axis=1)
    return best_i


def recall_at_10(a, np, dim, m):
    """queries/sec AND recall@10 on trained indexes over ONE embedding-like column (BASELINE.json's metric is
    "queries/sec @ recall@10"): `--recall-rows` x `dim` rows, two indexes —
      A: nlist 4096, m = 96 (C3's shape);  B: nlist = rows / 8192, m = dim / 16 (what the reference builds by default,
         table/create_index.rs:741-794, index/vector.rs:306-319) —
    each swept over nprobes x refine_factor with the bf16 raw column in HBM; every point carries its QPS (device-resident batches)
    and its recall@10 against (i) the engine's exact flat search, all queries, and (ii) an independent host truth (numpy sgemm)
    on the first 512; the CPU oracle answers 512 queries of three points per index row for row (ids must be ==).
    `qps_at_recall`: the fastest point of each index reaching recall 0.95 / 0.99."""
    import torch
    import lancedb_amd
    from lancedb_amd import _abi
    t0 = time.perf_counter()
    n, nq = a.recall_rows, a.recall_queries
    x, q, g, desc = recall_column(a, dim, "embedding")
    dev = x.device
    hq = q.cpu().numpy()
    fl = lancedb_amd.FlatIndex(x.contiguous())
    torch.cuda.synchronize()
    truth = fl.search(hq, k=10).rowids
    del fl
    torch.cuda.empty_cache()
    n_host = min(nq, 512)
    t1 = time.perf_counter()
    truth_host = host_truth_top10(np, x, hq[:n_host]) if a.cpu_seconds > 0 else None
    out = {"n_rows": n, "dim": dim, "queries": nq, "data": desc,
           "truth": "exact flat search over the f32 column (engine flat path), all queries; and numpy sgemm on the host for the first "
                    f"{n_host} (independent of the engine)",
           "raw_column": "bf16, index order, in HBM (the refine points re-rank on it)"}
    if truth_host is not None:
        agree = float(np.mean([len(set(truth[i].tolist()) & set(truth_host[i].tolist())) / 10.0 for i in range(n_host)]))
        out["engine_flat_truth_vs_host_truth_overlap"] = round(agree, 5)
        out["host_truth_seconds"] = round(time.perf_counter() - t1, 1)

    def rec(ids, tr, upto):
        return round(float(np.mean([len(set(tr[i].tolist()) & set(ids[i].tolist())) / 10.0 for i in range(upto)])), 4)
    B = min(a.batch, nq)
    qb = [q[i:i + B].contiguous() for i in range(0, nq - B + 1, B)][:4]
    dout = (torch.empty((B, 10), dtype=torch.int64, device=dev), torch.empty((B, 10), dtype=torch.float32, device=dev),
            torch.empty((B,), dtype=torch.int32, device=dev))
    small = n < 2_000_000  # (toy runs)
    # (on this column a query's neighbours sit in a handful of partitions: recall saturates at a few probes, and the 48-byte codes
    #  of index B need a deeper re-rank than A's 96-byte codes to reach the same recall)
    shapes = (("A", recall_nlist(a), m, (1, 4, 16, 64), (0, 5, 10)), ("B", max(8, n // 8192), dim // 16, (1, 5, 20), (0, 5, 10, 20)))
    points, n_or = [], min(nq, 512)
    for name, nlist, m_i, nprobes, rfs in shapes:
        nprobes = tuple(p_ for p_ in nprobes if p_ <= nlist)
        # the oracle answers three points per index: the widest probe list without and with refine, a narrower one with refine
        oracle_pts = ((nprobes[-1], 0), (nprobes[-1], 10), (nprobes[max(0, len(nprobes) - 2)], 5))
        R = train_and_encode(a, x, g, nlist, m_i)
        ix = lancedb_amd.IvfPqIndex(R["cen"].contiguous(), R["codebook"].contiguous(), R["part_offsets"], R["codes"], R["order"],
                                    raw_vectors=R["xs"].view(torch.int16), raw_dtype=_abi.DTYPE_BF16)
        ix.set_stream(torch.cuda.current_stream().cuda_stream)
        lens = np.diff(np.asarray(R["part_offsets"]).astype(np.int64))
        meta = {"nlist": nlist, "m": m_i, "train_seconds": round(R["t_train"], 2), "encode_rows_per_s": round(n / R["t_enc"]),
                "partition_rows_median": int(np.median(lens)), "partition_rows_max": int(lens.max())}
        ox = None
        if a.cpu_seconds > 0:
            from oracle import oracle as orc
            orc.build()
            ox = orc.OracleIndex(R["cen"].cpu().numpy(), R["codebook"].cpu().numpy(), R["part_offsets"], R["codes"].cpu().numpy(),
                                 R["order"].cpu().numpy().astype(np.uint64), raw_vectors=R["xs"].view(torch.int16).cpu().numpy().view(np.uint16),
                                 raw_dtype=_abi.DTYPE_BF16)
        for nprobe in nprobes:
            for rf in rfs:
                got = ix.search(hq, k=10, nprobe_min=nprobe, nprobe_max=nprobe, refine_factor=rf).rowids
                params = _abi.make_params(k=10, nprobe_min=nprobe, nprobe_max=nprobe, refine_factor=rf)
                for i in range(2):
                    ix.search(qb[i % len(qb)], params, out=dout)
                torch.cuda.synchronize()
                steps = 3 if small else max(4, a.steps // 2)
                t1 = time.perf_counter()
                for i in range(steps):
                    ix.search(qb[i % len(qb)], params, out=dout)
                torch.cuda.synchronize()
                qps = B * steps / (time.perf_counter() - t1)
                pt = {"index": name, "nlist": nlist, "m": m_i, "nprobe": nprobe, "refine_factor": rf, "batch_queries": B,
                      "queries_per_s": round(qps, 1), "recall_at_10": rec(got, truth, nq), "rows_scanned_per_query": int(ix.stats()["vectors_scanned"] // max(ix.stats()["n_queries"], 1))}
                if truth_host is not None:
                    pt["recall_at_10_vs_host_truth"] = rec(got, truth_host, n_host)
                if ox is not None and (nprobe, rf) in oracle_pts:
                    o_ids, _, _, _ = ox.search(hq[:n_or], k=10, nprobe_min=nprobe, nprobe_max=nprobe, refine_factor=rf)
                    pt["recall_at_10_cpu_oracle"] = rec(o_ids, truth, n_or)
                    pt["recall_at_10_engine_same_queries"] = rec(got, truth, n_or)
                    pt["rowids_bit_exact_vs_oracle"] = bool((o_ids == got[:n_or]).all())
                    out[f"{name}_nprobe{nprobe}_refine{rf}_rowids_bit_exact"] = pt["rowids_bit_exact_vs_oracle"]
                points.append(pt)
        at = {}
        for target in (0.95, 0.99):
            ok = [p_ for p_ in points if p_["index"] == name and p_["recall_at_10"] >= target]
            if ok:
                best = max(ok, key=lambda p_: p_["queries_per_s"])
                at[str(target)] = {k2: best[k2] for k2 in ("queries_per_s", "recall_at_10", "nprobe", "refine_factor")}
        meta["qps_at_recall"] = at
        out["index_" + name] = meta
        if ox is not None:
            ox.close()
        ix.close()
        del ix, R
        torch.cuda.empty_cache()
    out["points"] = points
    out["cpu_oracle_queries"] = n_or
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


def recall_embedding_like(a, np):
    """A second recall set whose neighbours are NOT near-ties ("1 M x 1536, unit-normalised for cosine"):
    unit vectors of intrinsic dimension 48 with a power-law spectrum (x = normalise(z W + 2 % noise), z from a
    2000-cluster mixture) — the shape of text-embedding columns — indexed with the reference's defaults for the
    dimension (m = dim / 16 = 96) and searched with cosine.  On the isotropic mixture of `recall_at_10` the true top-10
    of a query are separated by less than the PQ error, so nprobe does not matter and only refine moves recall; here
    both knobs do, which is what an nprobe / refine_factor sweep is for (BASELINE.md's chart: nprobes 25-100, rf 30-50)."""
    import torch
    import lancedb_amd
    t0 = time.perf_counter()
    n, nq, dim, m, nlist, idim = a.recall2_rows, a.recall2_queries, 1536, 96, 1024, 48
    dsub = dim // m
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(SEED + 11)
    spec = 1.0 / torch.sqrt(1.0 + torch.arange(idim, device=dev, dtype=torch.float32))
    W = torch.linalg.qr(torch.randn((dim, idim), generator=g, device=dev))[0].T.contiguous()  # [idim, dim], orthonormal rows
    centers = torch.randn((2000, idim), generator=g, device=dev)

    def draw(cnt):
        z = (centers[torch.randint(0, 2000, (cnt,), generator=g, device=dev)] + 0.35 * torch.randn((cnt, idim), generator=g, device=dev)) * spec
        v = z @ W + 0.02 * torch.randn((cnt, dim), generator=g, device=dev) * float(spec.norm()) / np.sqrt(dim)
        return v / v.norm(dim=1, keepdim=True)
    x = torch.empty((n, dim), device=dev)
    for r0 in range(0, n, 250_000):
        x[r0:r0 + 250_000] = draw(min(250_000, n - r0))
    q = draw(nq)
    iters = a.recall_iters
    pick = torch.randperm(n, generator=g, device=dev)
    ivf_rows = x[pick[:min(n, 256 * nlist)].sort().values].contiguous()
    init = ivf_rows[torch.randperm(ivf_rows.shape[0], generator=g, device=dev)[:nlist].sort().values].contiguous()
    torch.cuda.synchronize()
    cen, _ = lancedb_amd.kmeans_train(ivf_rows, init, metric="cosine", iters=iters)
    pq_rows = x[pick[:min(n, 256 * 256)].sort().values].contiguous()
    torch.cuda.synchronize()
    resid, _ = lancedb_amd.ivf_residuals(pq_rows, cen, metric="cosine")
    seeds = resid[torch.randperm(resid.shape[0], generator=g, device=dev)[:256].sort().values]
    cb0 = seeds.reshape(256, m, dsub).permute(1, 0, 2).contiguous()
    torch.cuda.synchronize()
    codebook = lancedb_amd.pq_train(resid, cb0, metric="cosine", iters=iters)
    torch.cuda.synchronize()
    del ivf_rows, pq_rows, resid
    part_offsets, codes, order = lancedb_amd.ivfpq_encode(x, cen, codebook, metric="cosine")
    xs = x[order].contiguous()
    torch.cuda.synchronize()
    ix = lancedb_amd.IvfPqIndex(cen.contiguous(), codebook.contiguous(), part_offsets, codes, order, raw_vectors=xs, metric="cosine")
    fl = lancedb_amd.FlatIndex(x.contiguous())
    hq = q.cpu().numpy()
    from lancedb_amd import _abi
    truth = fl.search(hq, k=10, metric=_abi.METRIC_COSINE).rowids
    del fl

    def rec(ids):
        return round(float(np.mean([len(set(truth[i].tolist()) & set(ids[i].tolist())) / 10.0 for i in range(nq)])), 4)
    out = {"n_rows": n, "dim": dim, "nlist": nlist, "m": m, "metric": "cosine", "queries": nq, "intrinsic_dim": idim,
           "truth": "exact flat cosine search (engine flat path)"}
    sweep = {}
    for nprobe in (1, 2, 4, 16, 64):
        for rf in (0, 5, 10, 25):
            got = ix.search(hq, k=10, nprobe_min=nprobe, nprobe_max=nprobe, refine_factor=rf)
            sweep[f"nprobe{nprobe}" + (f"_refine{rf}" if rf else "")] = rec(got.rowids)
    out["recall_at_10"] = sweep
    if a.cpu_seconds > 0:  # one operating point against the oracle, row for row
        from oracle import oracle as orc
        orc.build()
        ox = orc.OracleIndex(cen.cpu().numpy(), codebook.cpu().numpy(), part_offsets, codes.cpu().numpy(),
                             order.cpu().numpy().astype(np.uint64), raw_vectors=xs.cpu().numpy(), metric="cosine")
        sub = hq[:512]
        got = ix.search(sub, k=10, nprobe_min=16, nprobe_max=16, refine_factor=10)
        o_ids, o_d, _, _ = ox.search(sub, k=10, nprobe_min=16, nprobe_max=16, refine_factor=10)
        out["nprobe16_refine10_rowids_bit_exact_512_queries"] = bool((o_ids == got.rowids).all())
        out["nprobe16_refine10_distances_equal_512_queries"] = bool((o_d == got.distances).all())
        ox.close()
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


def traffic_source_from_profiles(workload, batch):
    """Which committed file `traffic` came from (it is NOT measured in the run it is printed in: PMC
    counters need their own rocprofv3 pass)."""
    import glob
    src = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")) + glob.glob(os.path.join(ROOT, "profiles", "*fetch_size*.json"))):
        try:
            t = json.load(open(path))
        except (OSError, ValueError):
            continue
        if t.get("workload") == workload and t.get("batch_queries") == batch:
            src = "profiles/" + os.path.basename(path) + " (committed rocprofv3 --pmc FETCH_SIZE x 2 pass of the same workload, not this run)"
    return src


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
    """The C oracle (a restatement, kind = "port") on this box's host cores over a bounded sample of the last batch;
    doubles as the full-size parity check.  `cores` = the threads it ran on: the process's affinity mask capped by the
    cgroup's CPU quota (bench_legs.host_cores — os.cpu_count() is the box, not what the container may use); the
    per-stage CPU seconds (summed over the threads) say where the oracle's time goes."""
    import bench_legs as legs
    from oracle import oracle as orc
    orc.build()
    hc = legs.host_cores()
    cores = hc["usable"]
    ox = orc.OracleIndex(centroids, codebook, part_offsets, codes, row_ids, metric="l2",
                         codes_layout=1, borrow=True)
    q = d_queries.cpu().numpy()
    n0 = min(len(q), cores)
    t0 = time.perf_counter()
    ids0, dist0, cnt0, st = ox.search(q[:n0], params, nthreads=cores)
    t_probe = time.perf_counter() - t0
    stage = dict(ox.last_stage_seconds)
    n_more = int(max(0, min(len(q) - n0, (a.cpu_seconds - t_probe) / max(t_probe, 1e-3) * cores)))
    n_more -= n_more % cores
    t1 = time.perf_counter()
    if n_more:
        ids1, dist1, cnt1, _ = ox.search(q[n0:n0 + n_more], params, nthreads=cores)
        for k2, v2 in ox.last_stage_seconds.items():
            stage[k2] += v2
    t_more = time.perf_counter() - t1
    nq = n0 + n_more
    ids = np.concatenate([ids0, ids1]) if n_more else ids0
    dst = np.concatenate([dist0, dist1]) if n_more else dist0
    g_ids = last[0][:nq].cpu().numpy().astype(np.uint64)
    g_dist = last[1][:nq].cpu().numpy()
    rel = float(np.max(np.abs(g_dist - dst) / np.maximum(np.abs(dst), 1e-30)))
    wall = t_probe + t_more
    busy = sum(stage.values())
    return {
        "value": nq / wall, "unit": "queries/s", "cores": cores, "host_cores": hc, "kind": "port",
        "queries_per_s_per_core": nq / wall / cores,
        "stage_cpu_seconds": {k2: round(v2, 3) for k2, v2 in stage.items()},
        "cpu_seconds_per_query": busy / max(nq, 1),
        "thread_utilisation": busy / max(wall * cores, 1e-9),
        "sample": f"{nq} queries of the last timed batch, one query per thread (OpenMP, {cores} threads = affinity {hc['affinity']}, "
                  f"cgroup CPU quota {hc['cgroup_cpu_quota']}, box {hc['box']}), {wall:.1f} s of wall time; C restatement of the "
                  "lance-index IVF-PQ path (oracle/ann_oracle.c, -O3 -mavx2 -mfma; coarse stage 8 centroids per AVX2 register, "
                  "ADC 8 sub-quantisers per sweep — same chains, same bits), not the reference binary",
        "parity": {"queries": nq, "rowids_bit_exact": bool((g_ids == ids).all()), "max_rel_distance_error": rel},
    }


if __name__ == "__main__":
    main()
