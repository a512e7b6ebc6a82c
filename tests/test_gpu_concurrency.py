"""Concurrent callers, latency mode and the device-side timeout of one index handle
(SURVEY.md §8b threading: callers are tokio workers, python/src/runtime.rs:31-37; BaseTable is
Send + Sync, table.rs:549; QueryExecutionOptions.timeout, query.rs:626-658, utils/mod.rs:328-392)."""
import threading
import time

import numpy as np
import pytest

import lancedb_amd
from lancedb_amd import _abi
from oracle import train

pytestmark = pytest.mark.gpu


def test_sixty_four_threads_single_queries_are_coalesced_and_exact(oracle):
    s = train.synthetic_index(120000, 128, 64, 32, seed=4, skew=0.8)
    ix = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    rng = np.random.default_rng(1)
    n_threads, per_thread = 64, 12
    qs = rng.normal(size=(n_threads, per_thread, 128)).astype(np.float32)
    # two parameter sets in flight at once: only calls with equal parameters share a device batch
    kws = [dict(k=10, nprobe_min=16, nprobe_max=16), dict(k=25, nprobe_min=8, nprobe_max=8)]
    exp = {}
    for t in range(n_threads):
        exp[t] = [o.search(qs[t, i:i + 1], **kws[t % 2]) for i in range(per_thread)]
    errors, coalesced = [], []
    barrier = threading.Barrier(n_threads)

    def worker(t):
        try:
            barrier.wait()
            for i in range(per_thread):
                got = ix.search(qs[t, i:i + 1], **kws[t % 2])
                ids, dist, cnt, _ = exp[t][i]
                assert (got.counts == cnt).all() and (got.rowids == ids).all() and (got.distances == dist).all()
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    t0 = time.perf_counter()
    [t.start() for t in threads]
    [t.join() for t in threads]
    wall = time.perf_counter() - t0
    assert not errors, errors[:3]
    # the same calls one after another on one thread, for the record (no assertion on speed)
    t1 = time.perf_counter()
    for i in range(per_thread):
        ix.search(qs[0, i:i + 1], **kws[0])
    serial = (time.perf_counter() - t1) / per_thread
    print(f"64 threads x {per_thread} single-query calls: {n_threads * per_thread / wall:.0f} QPS; "
          f"one thread: {1 / serial:.0f} QPS")


def test_graph_replay_serves_small_batches_and_stays_exact(oracle):
    s = train.synthetic_index(60000, 128, 32, 32, seed=5)
    raw = np.random.default_rng(3).normal(size=(60000, 128)).astype(np.float32)
    ix = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw)
    rng = np.random.default_rng(2)
    ix.configure(graph=True)  # opt in: replay is off after open (it measured slower than eager launches)
    for kw in (dict(k=10, nprobe_min=8, nprobe_max=8), dict(k=10, nprobe_min=8, nprobe_max=8, refine_factor=5),
               dict(k=10, nprobe_min=8, nprobe_max=8, upper_bound=30.0)):
        for rep in range(5):  # 1st eager (sizes the workspace), 2nd captures, 3rd.. replay
            q = rng.normal(size=(1, 128)).astype(np.float32)
            got = ix.search(q, **kw)
            ids, dist, cnt, _ = o.search(q, **kw)
            assert (got.counts == cnt).all() and (got.rowids == ids).all() and (got.distances == dist).all(), (kw, rep)
    replays = ix.stats()["graph_replays"]
    assert replays >= 6, replays  # >= 2 replays per parameter set (the stats survive the per-call reset)
    # a different batch size re-sizes the workspace: stale graphs are re-captured, results stay exact
    q = rng.normal(size=(48, 128)).astype(np.float32)
    for _ in range(3):
        got = ix.search(q, k=10, nprobe_min=8, nprobe_max=8)
        ids, dist, cnt, _ = o.search(q, k=10, nprobe_min=8, nprobe_max=8)
        assert (got.rowids == ids).all() and (got.distances == dist).all()
    # graph / coalescing off: the eager path
    ix.configure(graph=False, coalesce=False)
    q1 = rng.normal(size=(1, 128)).astype(np.float32)
    before = ix.stats()["graph_replays"]
    for _ in range(3):
        got = ix.search(q1, k=10, nprobe_min=8, nprobe_max=8)
    assert (got.rowids == o.search(q1, k=10, nprobe_min=8, nprobe_max=8)[0]).all()
    assert ix.stats()["graph_replays"] == 0 and before >= 0


def test_timeout_stops_the_scan_on_the_device(oracle):
    """A deadline far shorter than the scan: the persistent scan stops popping work items and the
    call returns Timeout (host I/O) / reports it at sync (device I/O); the handle stays usable."""
    s = train.synthetic_index(3_000_000, 128, 64, 32, seed=8, skew=0.3)
    ix = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    q = np.random.default_rng(1).normal(size=(4096, 128)).astype(np.float32)
    t0 = time.perf_counter()
    full = ix.search(q, k=10, nprobe_min=64, nprobe_max=64)
    t_full = time.perf_counter() - t0
    assert ix.stats()["timed_out"] == 0
    t0 = time.perf_counter()
    full = ix.search(q, k=10, nprobe_min=64, nprobe_max=64)  # warm (workspace sized)
    t_full = time.perf_counter() - t0
    t0 = time.perf_counter()
    with pytest.raises(lancedb_amd.QueryTimeout, match="stopped on the device"):
        ix.search(q, k=10, nprobe_min=64, nprobe_max=64, timeout_ms=1)
    t_cut = time.perf_counter() - t0
    assert ix.stats()["timed_out"] == 1
    print(f"full search {t_full * 1e3:.1f} ms, with a 1 ms deadline {t_cut * 1e3:.1f} ms")
    assert t_cut < 0.6 * t_full, "the scan ran to completion"
    # device I/O: the call returns at once, the timeout surfaces at sync
    DA = lancedb_amd.DeviceArray
    dq = DA.from_numpy(q)
    out = (DA((4096, 10), np.int64), DA((4096, 10), np.float32), DA((4096,), np.int32))
    ix.search(dq, _abi.make_params(k=10, nprobe_min=64, nprobe_max=64, timeout_ms=1), out=out)
    with pytest.raises(lancedb_amd.QueryTimeout):
        ix.sync()
    # and the handle is fine afterwards: a generous deadline returns the full result
    again = ix.search(q, k=10, nprobe_min=64, nprobe_max=64, timeout_ms=max(60_000, int(t_full * 20_000)))
    assert (again.rowids == full.rowids).all() and (again.distances == full.distances).all()
    # the generic scan kernel honours the deadline too
    s2 = train.synthetic_index(2_000_000, 32, 32, 8, seed=9, skew=0.3)
    ix2 = lancedb_amd.IvfPqIndex(s2["centroids"], s2["codebook"], s2["part_offsets"], s2["codes"], s2["row_ids"])
    q2 = np.random.default_rng(2).normal(size=(2048, 32)).astype(np.float32)
    with pytest.raises(lancedb_amd.QueryTimeout):
        ix2.search(q2, k=10, nprobe_min=32, nprobe_max=32, timeout_ms=1)


def test_refine_from_host_mapped_raw_vectors(oracle):
    """MI355_INDEX_RAW_HOST_MAPPED (C5: a raw column that does not fit HBM): the refine stage gathers
    its k * refine_factor rows per query from the caller's page-locked host memory; also on shard
    handles, whose local positions are converted to global index positions."""
    rng = np.random.default_rng(12)
    for m, dim in ((8, 64), (32, 128)):
        s = train.synthetic_index(50000, dim, 40, m, seed=m, skew=0.8, empty_parts=3)
        for dt, code in ((np.float32, _abi.DTYPE_F32), (np.float16, _abi.DTYPE_F16)):
            raw = rng.normal(size=(50000, dim)).astype(dt)
            raw_bits = raw if dt is np.float32 else raw.view(np.uint16)
            o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"],
                                   raw_vectors=raw_bits, metric="cosine", raw_dtype=code)
            g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"],
                                       raw_vectors=raw_bits, metric="cosine", raw_dtype=code, raw_host_mapped=True)
            q = rng.normal(size=(24, dim)).astype(np.float32)
            for kw in (dict(k=10, nprobe_min=16, nprobe_max=16, refine_factor=10), dict(k=10, nprobe_min=4, nprobe_max=4, refine_factor=50)):
                got = g.search(q, **kw)
                ids, dist, cnt, _ = o.search(q, **kw)
                assert (got.counts == cnt).all() and (got.rowids == ids).all() and (got.distances == dist).all()
            g.close()
    # a shard handle addresses the caller's column by GLOBAL index position: its own search (local
    # refine of its local candidates) equals the oracle over the rows the shard keeps
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from sharded_model import shard_local
    s = train.synthetic_index(30000, 64, 24, 8, seed=77, skew=0.9, empty_parts=2)
    raw = rng.normal(size=(30000, 64)).astype(np.float32)
    s["raw"] = raw
    owner = lancedb_amd.shard_plan(s["part_offsets"], 3)
    q = rng.normal(size=(16, 64)).astype(np.float32)
    for r in range(3):
        loc = shard_local(s, owner, r)
        o = oracle.OracleIndex(loc["centroids"], loc["codebook"], loc["part_offsets"], loc["codes"], loc["row_ids"], raw_vectors=loc["raw"])
        g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw,
                                   raw_host_mapped=True, shard_count=3, shard_rank=r)
        got = g.search(q, k=10, nprobe_min=12, nprobe_max=12, refine_factor=6)
        ids, dist, cnt, _ = o.search(q, k=10, nprobe_min=12, nprobe_max=12, refine_factor=6)
        assert (got.counts == cnt).all() and (got.rowids == ids).all() and (got.distances == dist).all()


@pytest.mark.parametrize("metric", ["l2", "cosine", "dot"])
def test_host_column_refine_for_every_dtype_and_ragged_row_length(oracle, metric):
    """Refine over a host-mapped column (MI355_INDEX_RAW_HOST_MAPPED) in all three element types: rows whose byte length is a
    multiple of 16 (dim 72 f32 = 288 B, dim 40 bf16 = 80 B) take 16-B pieces, the others (dim 50 f16 = 100 B) scalar loads;
    candidate counts that do not fill a wave (k * refine_factor = 77, 130) — distances must be exactly the oracle's."""
    rng = np.random.default_rng(5)
    for dim, m in ((72, 9), (40, 8), (50, 10), (256, 32)):
        n = 40000
        s = train.synthetic_index(n, dim, 24, m, seed=dim, skew=0.6, empty_parts=1)
        rawf = rng.normal(size=(n, dim)).astype(np.float32)
        cols = [(rawf, _abi.DTYPE_F32), (rawf.astype(np.float16).view(np.uint16), _abi.DTYPE_F16)]
        # bf16 bits: round-to-nearest-even of the f32 pattern
        u = rawf.view(np.uint32).astype(np.uint64)
        bf = (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)
        cols.append((bf, _abi.DTYPE_BF16))
        for raw_bits, code in cols:
            o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"],
                                   raw_vectors=raw_bits, metric=metric, raw_dtype=code)
            g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"],
                                       raw_vectors=raw_bits, metric=metric, raw_dtype=code, raw_host_mapped=True)
            q = rng.normal(size=(5, dim)).astype(np.float32)
            for kw in (dict(k=7, refine_factor=11), dict(k=10, refine_factor=13), dict(k=1, refine_factor=3)):
                kw = dict(nprobe_min=8, nprobe_max=8, **kw)
                got = g.search(q, **kw)
                ids, dist, cnt, _ = o.search(q, **kw)
                assert (got.counts == cnt).all() and (got.rowids == ids).all() and (got.distances == dist).all(), (dim, code, kw)
            g.close()


def test_deferred_refine_overlaps_the_next_call_and_stays_exact(oracle):
    """Device-I/O refine calls leave their exact re-rank on the handle's refine stream (two buffer sets): several
    calls in flight, results complete at sync(), every batch == the oracle; other entry points join first."""
    rng = np.random.default_rng(12)
    n, dim, m = 120000, 128, 32
    s = train.synthetic_index(n, dim, 48, m, seed=6, skew=0.7)
    raw = rng.normal(size=(n, dim)).astype(np.float32)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw)
    DA = lancedb_amd.DeviceArray
    for host_mapped in (False, True):
        ix = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw,
                                    raw_host_mapped=host_mapped)
        ix.configure(defer_refine=True)  # opt-in (MI355_CFG_DEFER_REFINE): only the host-mapped column actually defers
        qs = [rng.normal(size=(40, dim)).astype(np.float32) for _ in range(6)]
        dq = [DA.from_numpy(q) for q in qs]
        outs = [(DA((40, 10), np.int64), DA((40, 10), np.float32), DA((40,), np.int32)) for _ in qs]
        p = _abi.make_params(k=10, nprobe_min=12, nprobe_max=12, refine_factor=8)
        res = [ix.search(dq[i], p, out=outs[i]) for i in range(len(qs))]  # no sync in between
        ix.sync()
        for q, r in zip(qs, res):
            ids, dist, cnt, _ = o.search(q, k=10, nprobe_min=12, nprobe_max=12, refine_factor=8)
            assert (r.rowids.numpy().view(np.uint64) == ids).all() and (r.distances.numpy() == dist).all() and (r.counts.numpy() == cnt).all()
        # a deferred call followed at once by a host-I/O call and by stats(): both join the pending re-rank
        r = ix.search(dq[0], p, out=outs[0])
        h = ix.search(qs[1], k=10, nprobe_min=12, nprobe_max=12)
        assert (h.rowids == o.search(qs[1], k=10, nprobe_min=12, nprobe_max=12)[0]).all()
        assert ix.stats()["n_queries"] >= 40
        ix.sync()
        assert (r.rowids.numpy().view(np.uint64) == o.search(qs[0], k=10, nprobe_min=12, nprobe_max=12, refine_factor=8)[0]).all()


def test_deferred_refine_with_changing_batch_sizes_and_refine_factors(oracle):
    """back-to-back deferred calls of DIFFERENT shapes.  The two buffer sets used to be carved
    out of one allocation at offsets that depended on the current call's shape, so a smaller call's ANN list landed in
    the range the previous call's re-rank was still reading.  Each set now has its own allocation; no sync in between."""
    rng = np.random.default_rng(3)
    n, dim, m = 90000, 128, 32
    s = train.synthetic_index(n, dim, 32, m, seed=8, skew=0.7)
    raw = rng.normal(size=(n, dim)).astype(np.float32)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw)
    DA = lancedb_amd.DeviceArray
    ix = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], raw_vectors=raw,
                                raw_host_mapped=True)
    ix.configure(defer_refine=True)
    shapes = [(100, 10, 8), (50, 10, 8), (7, 10, 30), (120, 5, 3), (1, 10, 50), (64, 20, 8), (100, 10, 8), (3, 1, 100)]
    for rounds in range(2):
        calls = []
        for nq, k, rf in shapes:
            q = rng.normal(size=(nq, dim)).astype(np.float32)
            out = (DA((nq, k), np.int64), DA((nq, k), np.float32), DA((nq,), np.int32))
            p = _abi.make_params(k=k, nprobe_min=10, nprobe_max=10, refine_factor=rf)
            calls.append((q, DA.from_numpy(q), out, p, k, rf))
        res = [ix.search(c[1], c[3], out=c[2]) for c in calls]  # no sync in between
        ix.sync()
        for (q, _, _, _, k, rf), r in zip(calls, res):
            ids, dist, cnt, _ = o.search(q, k=k, nprobe_min=10, nprobe_max=10, refine_factor=rf)
            assert (r.counts.numpy() == cnt).all() and (r.rowids.numpy().view(np.uint64) == ids).all() and (r.distances.numpy() == dist).all()
    # without the opt-in the results of a device-I/O call are ordered on the handle's stream: a plain stream sync suffices
    ix.configure(defer_refine=False)
    q, dq, out, p, k, rf = calls[0]
    r = ix.search(dq, p, out=out)
    from lancedb_amd import _hip
    _hip.runtime().hipDeviceSynchronize()
    ids, dist, cnt, _ = o.search(q, k=k, nprobe_min=10, nprobe_max=10, refine_factor=rf)
    assert (r.rowids.numpy().view(np.uint64) == ids).all()


@pytest.mark.timeout(300, method="thread")  # (a lost wake-up would block inside the C call: end the process, do not hang the box)
def test_two_hundred_native_threads_each_get_their_own_results():
    """The coalescing queue under the reference's kind of caller (OS threads, no interpreter lock; tests/tools/loadgen.cpp):
    200 threads x 16 single-query calls must return, call by call, what one thread gets for the same queries — the load
    generator sums the row ids every call returned.  Parked callers are released by futex word, the device is handed from
    leader to leader: a lost wake-up hangs this test (pytest's timeout), a mixed-up result changes the sum."""
    import ctypes as C

    import bench_legs as legs
    from lancedb_amd._lib import lib

    s = train.synthetic_index(150000, 64, 128, 16, seed=9, skew=0.6)
    ix = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    rng = np.random.default_rng(5)
    pool = 512
    hq = np.ascontiguousarray(rng.normal(size=(pool, 64)), dtype=np.float32)
    params = _abi.make_params(k=10, nprobe_min=8, nprobe_max=8)
    params.io_mem = _abi.MEM_HOST
    L = C.CDLL(legs.build_loadgen())
    L.loadgen_run.restype = C.c_int32
    fn = C.cast(lib().mi355_search, C.c_void_p)

    def run(threads, per):
        sec, chk = C.c_double(0), C.c_uint64(0)
        st = L.loadgen_run(fn, ix._h, hq.ctypes.data_as(C.c_void_p), C.c_uint32(pool), C.c_uint32(64), C.byref(params), C.c_uint32(10),
                           C.c_uint32(threads), C.c_uint32(per), C.byref(sec), None, C.byref(chk))
        assert st == 0
        return chk.value, threads * per / sec.value

    one, _ = run(1, 3200)
    for threads, per in ((200, 16), (64, 50), (8, 400)):
        got, qps = run(threads, per)
        assert got == one, (threads, got, one)
        print(f"{threads} native threads: {qps:.0f} QPS")


def test_a_timeout_is_a_calls_own_in_a_coalesced_batch(oracle):
    """QueryExecutionOptions.timeout is per query (rust/lancedb/src/query.rs:641).  Round 5 armed a coalesced batch with its OLDEST
    call's deadline: one call that had expired while parked failed the whole batch — newcomers included — and, the queue being
    collected front to back, every batch after it.  Now an expired call gets status 3 alone.  48 threads hammer a
    handle with a budget a parked call sometimes misses: every call must end OK with exact results or with Timeout, and the OK
    calls must be the bulk — with the cascade they were almost none."""
    s = train.synthetic_index(400000, 128, 64, 32, seed=6, skew=0.5)
    ix = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    rng = np.random.default_rng(2)
    n_threads, per_thread = 48, 30
    qs = rng.normal(size=(n_threads, 128)).astype(np.float32)
    kw = dict(k=10, nprobe_min=64, nprobe_max=64)
    exp = [o.search(qs[t:t + 1], **kw) for t in range(n_threads)]
    # how long one coalesced round of everybody takes on this box: the budget is about one and a half such rounds
    t0 = time.perf_counter()
    for _ in range(5):
        ix.search(qs, **kw)
    round_ms = (time.perf_counter() - t0) / 5 * 1e3
    budget = max(2, int(round_ms * 1.5 + 1))
    ok, late, errors = [0] * n_threads, [0] * n_threads, []
    barrier = threading.Barrier(n_threads)

    def worker(t):
        try:
            barrier.wait()
            for i in range(per_thread):
                try:
                    got = ix.search(qs[t:t + 1], timeout_ms=budget, **kw)
                except lancedb_amd.QueryTimeout:
                    late[t] += 1
                    continue
                ids, dist, cnt, _ = exp[t]
                assert (got.counts == cnt).all() and (got.rowids == ids).all() and (got.distances == dist).all()
                ok[t] += 1
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors[:3]
    total = n_threads * per_thread
    assert sum(ok) + sum(late) == total
    print(f"budget {budget} ms (one round of {n_threads} callers: {round_ms:.2f} ms): {sum(ok)} ok, {sum(late)} timed out")
    assert sum(ok) >= total // 2, (sum(ok), sum(late))
    # and the handle is healthy afterwards
    got = ix.search(qs[:3], **kw)
    e = o.search(qs[:3], **kw)
    assert (got.rowids == e[0]).all() and (got.distances == e[1]).all()
