"""Throughput of mi355_ivfpq_encode (index population) at the C3 index shape
(dim 768, nlist 4096, m 96) on device-resident rows, with the CPU oracle
(OpenMP, all host cores) timed on a sample.  python -u tests/tools/bench_encode.py [rows]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import lancedb_amd  # noqa: E402
from lancedb_amd import DeviceArray  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
    dim, nlist, m = 768, 4096, 96
    rng = np.random.default_rng(0)
    cent = rng.standard_normal((nlist, dim), dtype=np.float32)
    cb = rng.standard_normal((m, 256, dim // m), dtype=np.float32) * np.float32(0.5)
    x = cent[rng.integers(0, nlist, size=n)] + rng.standard_normal((n, dim), dtype=np.float32) * np.float32(0.5)
    dx, dc, dcb = DeviceArray.from_numpy(x), DeviceArray.from_numpy(cent), DeviceArray.from_numpy(cb)
    lancedb_amd.ivfpq_encode(dx, dc, dcb)  # warm-up (allocations, code objects)
    lancedb_amd.synchronize()
    t0 = time.perf_counter()
    po, codes, order = lancedb_amd.ivfpq_encode(dx, dc, dcb)
    lancedb_amd.synchronize()
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    hpo, hcodes, horder = lancedb_amd.ivfpq_encode(x, cent, cb)
    t_host = time.perf_counter() - t0
    assert (hpo == po).all() and (hcodes == codes.numpy()).all()
    out = dict(rows=n, dim=dim, nlist=nlist, m=m, device_resident_rows_per_s=n / t_dev, seconds=t_dev,
               host_io_rows_per_s=n / t_host)
    try:
        from oracle import oracle
        ns = min(n, 20000)
        t0 = time.perf_counter()
        epo, ecodes, eorder, _ = oracle.ivfpq_encode(x[:ns], cent, cb, "l2")
        out["oracle_rows_per_s"] = ns / (time.perf_counter() - t0)
        got = lancedb_amd.ivfpq_encode(x[:ns], cent, cb)
        out["sample_bit_exact"] = bool((got[0] == epo).all() and (got[1] == ecodes).all() and (got[2] == eorder).all())
    except Exception as e:  # noqa: BLE001
        out["oracle_error"] = repr(e)
    # training: Lloyd iterations of the IVF k-means on a sample (sample_rate 64 -> 262144 rows)
    ns, iters = min(n, 262144), 5
    init = DeviceArray.from_numpy(x[rng.choice(ns, size=nlist, replace=False)])
    ds = DeviceArray.from_numpy(x[:ns])
    lancedb_amd.kmeans_train(ds, init, iters=1)
    lancedb_amd.synchronize()
    t0 = time.perf_counter()
    lancedb_amd.kmeans_train(ds, init, iters=iters)
    lancedb_amd.synchronize()
    t_km = time.perf_counter() - t0
    out["kmeans"] = dict(rows=ns, k=nlist, iters=iters, seconds=t_km, row_iters_per_s=ns * iters / t_km)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
