"""Seeded random-configuration parity sweeps and host-contract checks of the HIP
path (through the C ABI) against the CPU oracle: shapes, partition-length
profiles, k, nprobe ranges, metrics, filters, ranges and refine drawn at random
so that combinations no hand-written case names still get compared bit for bit."""
import threading

import numpy as np
import pytest

import lancedb_amd
from lancedb_amd import _abi
from oracle import train

pytestmark = pytest.mark.gpu


def _same(got, exp):
    ids, dist, cnt, st = exp
    assert st == 0
    assert (got.counts == cnt).all() and (got.rowids == ids).all() and (got.distances == dist).all()


def _random_lens(rng, nlist, n):
    kind = rng.integers(0, 4)
    if kind == 0:  # balanced
        w = np.ones(nlist)
    elif kind == 1:  # log-normal skew
        w = np.exp(rng.normal(0, 1.0, nlist))
    elif kind == 2:  # a few giants, many empties
        w = np.zeros(nlist)
        w[rng.choice(nlist, size=max(1, nlist // 6), replace=False)] = rng.random(max(1, nlist // 6)) + 0.1
    else:  # tiny partitions: fewer rows than a tile
        w = rng.random(nlist)
        n = min(n, nlist * 40)
    lens = rng.multinomial(n, w / w.sum())
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)


@pytest.mark.parametrize("seed", range(24))
def test_ivfpq_random_configuration(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    m = int(rng.choice([4, 8, 16, 32, 48, 64, 80, 96]))
    dsub = int(rng.choice([1, 2, 4, 8]))
    dim = m * dsub
    nlist = int(rng.integers(1, 40))
    n = int(rng.integers(1, 60000))
    metric = str(rng.choice(["l2", "cosine", "dot"]))
    s = train.synthetic_index(n, dim, nlist, m, seed=seed)
    s["part_offsets"] = _random_lens(rng, nlist, n)
    n = int(s["part_offsets"][-1])
    s["codes"], s["row_ids"] = s["codes"][:n], s["row_ids"][:n]
    if rng.random() < 0.3:  # collisions: many identical codes -> tie-breaks by row id
        s["codes"][: n // 2] = s["codes"][0] if n else 0
    raw = rng.normal(size=(n, dim)).astype(np.float32) if rng.random() < 0.5 else None
    layout = _abi.CODES_ROW_MAJOR
    codes = s["codes"]
    if rng.random() < 0.5:
        layout, codes = _abi.CODES_PART_TRANSPOSED, train.to_part_transposed(s["codes"], s["part_offsets"])
    g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], codes, s["row_ids"],
                               raw_vectors=raw, metric=metric, codes_layout=layout)
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], codes, s["row_ids"],
                           raw_vectors=raw, metric=metric, codes_layout=layout)
    nq = int(rng.integers(1, 70))
    q = (s["centroids"][rng.integers(0, nlist, size=nq)] + rng.normal(0, 0.7, size=(nq, dim))).astype(np.float32)
    for _ in range(4):
        k = int(rng.choice([1, 3, 10, 33, 64, 65, 128, 200]))
        np_min = int(rng.integers(1, nlist + 3))
        np_max = rng.choice([np_min, np_min + int(rng.integers(0, nlist)), None])
        kw = dict(k=k, nprobe_min=np_min, nprobe_max=None if np_max is None else int(np_max))
        if raw is not None and rng.random() < 0.5 and k * 4 <= 256:
            kw["refine_factor"] = int(rng.choice([1, 2, 4]))
        if rng.random() < 0.3 and n:
            sel = rng.choice(s["row_ids"], size=int(rng.integers(1, n + 1)), replace=False)
            kw["allow_rowids" if rng.random() < 0.5 else "block_rowids"] = sel
        if rng.random() < 0.3:
            ref = o.search(q, **{**kw, "k": min(64, max(k, 8))})[1]
            fin = ref[np.isfinite(ref)]
            if fin.size > 4:
                lo, hi = np.quantile(fin, [0.2, 0.8])
                if rng.random() < 0.5:
                    kw["lower_bound"] = float(lo)
                kw["upper_bound"] = float(hi)
        _same(g.search(q, **kw), o.search(q, **kw))


@pytest.mark.parametrize("seed", range(8))
def test_flat_random_configuration(oracle, seed):
    rng = np.random.default_rng(2000 + seed)
    n = int(rng.choice([1, 17, 4095, 4096, 4097, 20000, 70001]))
    dim = int(rng.choice([3, 8, 48, 64, 100, 128, 768]))
    dt = int(rng.choice([_abi.DTYPE_F32, _abi.DTYPE_BF16, _abi.DTYPE_F16]))
    v32 = (rng.normal(size=(n, dim)) * rng.choice([1e-3, 1.0, 50.0])).astype(np.float32)
    if dt == _abi.DTYPE_F32:
        v = v32
    elif dt == _abi.DTYPE_BF16:
        v = (v32.view(np.uint32) >> 16).astype(np.uint16)
    else:
        v = v32.astype(np.float16).view(np.uint16)
    rid = rng.permutation(n).astype(np.uint64) + (1 << 33) if rng.random() < 0.5 else None
    f = lancedb_amd.FlatIndex(v, rid, dtype=dt)
    nq = int(rng.integers(1, 140))
    q = rng.normal(size=(nq, dim)).astype(np.float32)
    for metric in ("l2", "cosine", "dot"):
        mt = _abi.METRIC_NAMES[metric]
        k = int(rng.choice([1, 10, 64, 130]))
        kw = dict(k=k, metric=mt)
        if rng.random() < 0.3:
            ids = rid if rid is not None else np.arange(n, dtype=np.uint64)
            kw["block_rowids"] = rng.choice(ids, size=int(rng.integers(1, n + 1)), replace=False)
        _same(f.search(q, **kw), oracle.flat_search(v, q, row_ids=rid, dtype=dt, **kw))


def test_concurrent_searches_on_one_handle(oracle):
    """BaseTable is Send + Sync and callers are tokio worker threads
    (python/src/runtime.rs:31-37): concurrent calls on one handle must serialise
    correctly (ctypes releases the GIL during the call)."""
    s = train.synthetic_index(60000, 128, 32, 32, seed=3)
    g = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    o = oracle.OracleIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"])
    rng = np.random.default_rng(0)
    jobs = [(rng.normal(size=(int(rng.integers(1, 50)), 128)).astype(np.float32), int(rng.choice([1, 10, 100])),
             int(rng.integers(1, 33))) for _ in range(24)]
    exp = [o.search(q, k=k, nprobe_min=npb, nprobe_max=npb) for q, k, npb in jobs]
    got = [None] * len(jobs)

    def work(lo):
        for i in range(lo, len(jobs), 4):
            q, k, npb = jobs[i]
            got[i] = g.search(q, k=k, nprobe_min=npb, nprobe_max=npb)

    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    for r, e in zip(got, exp):
        _same(r, e)


def test_timeout_maps_to_query_timeout(oracle):
    """QueryExecutionOptions.timeout (query.rs:641) -> Error::Timeout."""
    v = np.random.default_rng(0).normal(size=(1000000, 64)).astype(np.float32)
    f = lancedb_amd.FlatIndex(v)
    q = np.random.default_rng(1).normal(size=(512, 64)).astype(np.float32)
    with pytest.raises(lancedb_amd.QueryTimeout, match="Query timeout"):
        # lower-bounded range -> exact sweep of 512 x 1M rows: far longer than 1 ms
        f.search(q, k=10, lower_bound=0.0, timeout_ms=1)
    assert f.search(q[:8], k=10, timeout_ms=60000).counts.min() == 10
