"""Dev: the C3 workload at other PQ shapes (bench_legs.width_lines) with a checksum of the returned row ids, for A/B runs.
usage: [MI355_ANN_LIB=...] python tests/tools/width_time.py rows dim:m[:nbits] ..."""
import sys
import types

import numpy as np
import torch

sys.path.insert(0, ".")
import bench_legs as legs  # noqa: E402

rows = int(sys.argv[1])
shapes = []
for spec in sys.argv[2:]:
    p = [int(x) for x in spec.split(":")]
    shapes.append((p[0], p[1], p[2] if len(p) > 2 else 8))
a = types.SimpleNamespace(nlist=4096, nprobe=64, k=10, batch=2048, skew=0.5, steps=9)
dev = torch.device("cuda", 0)
out = legs.width_lines(a, torch, np, dev, shapes=tuple(shapes), n_rows=rows)
for name, line in out.items():
    print(f"{name}: {line['value']:.0f} QPS, scan {line['stage_us_per_step']['scan']:.0f} us, frac {line['roofline']['frac']:.3f}, checksum {line['rowid_checksum']}", flush=True)
