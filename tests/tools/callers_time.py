"""Dev: throughput / latency of concurrent single-query host callers (C++ threads, tests/tools/loadgen.cpp) on a C3-shaped
index.  usage: [MI355_ANN_LIB=...] python tests/tools/callers_time.py [rows]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import bench_legs as legs  # noqa: E402
import lancedb_amd  # noqa: E402
from lancedb_amd import _abi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
dim, m, nprobe = 768, 96, 64
nlist = max(64, n // 24_414)
dev = torch.device("cuda", 0)
s = legs.synth_ivfpq(torch, np, dev, n, dim, nlist, m, 0.5)
ix = lancedb_amd.IvfPqIndex(s["centroids"], s["codebook"], s["part_offsets"], s["codes"], s["row_ids"], codes_layout=_abi.CODES_PART_TRANSPOSED)
del s["codes"]
hq = legs.query_pool(torch, s, nlist, dim, 2048, 1)[0].cpu().numpy()
res = legs.concurrent_callers(np, ix, hq, _abi.make_params(k=10, nprobe_min=nprobe, nprobe_max=nprobe), 10, thread_counts=(1, 8, 32, 64, 128, 256))
for k, v in res.items():
    print(f"{k}: {v['queries_per_s']:.0f} QPS, p50 {v['latency_us_p50']:.0f} us, p99 {v['latency_us_p99']:.0f} us", flush=True)
