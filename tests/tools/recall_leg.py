"""Dev: the bench's QPS-at-recall leg alone (bench.recall_at_10: one embedding-like column, two trained indexes, nprobes x refine
sweep with QPS, recall against the engine's flat truth and an independent host truth, oracle parity on three points per index).
usage: python tests/tools/recall_leg.py [rows queries iters]"""
import json
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

a = types.SimpleNamespace(recall_rows=int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000,
                          recall_queries=int(sys.argv[2]) if len(sys.argv) > 2 else 4096,
                          recall_iters=int(sys.argv[3]) if len(sys.argv) > 3 else 25, batch=2048, steps=10, cpu_seconds=15.0)
r = bench.recall_at_10(a, np, 768, 96)
pts = r.pop("points")
print(json.dumps(r, indent=1))
for p in pts:
    print({k: p[k] for k in ("index", "nprobe", "refine_factor", "queries_per_s", "recall_at_10", "recall_at_10_vs_host_truth", "rows_scanned_per_query") if k in p},
          {k: p[k] for k in ("rowids_bit_exact_vs_oracle",) if k in p})
