"""Dev: the flat GEMM kernel's own time (mi355_flat_last_stats) for one library, several schedules.
usage: [MI355_ANN_LIB=...] python tests/tools/flat_gemm_time.py rows variant:grid:metric [variant:grid:metric ...]
(the legacy form `rows variant [grid] [metric]` still works).  Every configuration also reports the
queries that fell back to the exact sweep and a census of the filter matrix (never-filter / non-finite
entries), so a schedule that computes garbage cannot pass as merely slow."""
import sys
import time

import torch

sys.path.insert(0, ".")
import lancedb_amd  # noqa: E402
from lancedb_amd import _abi  # noqa: E402

rows = int(sys.argv[1])
if ":" in sys.argv[2]:
    cfgs = [(int(a), int(b), c) for a, b, c in (x.split(":") for x in sys.argv[2:])]
else:
    cfgs = [(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 0, sys.argv[4] if len(sys.argv) > 4 else "l2")]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(0x1A2CE)
col = torch.empty((rows, 768), device=dev, dtype=torch.bfloat16)
for r0 in range(0, rows, 2_000_000):
    col[r0:r0 + 2_000_000] = torch.randn((min(2_000_000, rows - r0), 768), generator=g, device=dev).to(torch.bfloat16)
q = torch.randn((1024, 768), generator=g, device=dev)
# the handle works on its own stream: the column must be complete before it is handed over
torch.cuda.synchronize()
fl = lancedb_amd.FlatIndex(col.view(torch.int16), dtype=_abi.DTYPE_BF16)
out = (torch.empty((1024, 10), dtype=torch.int64, device=dev), torch.empty((1024, 10), dtype=torch.float32, device=dev),
       torch.empty((1024,), dtype=torch.int32, device=dev))
for variant, grid, metric in cfgs:
    p = _abi.make_params(k=10, nprobe_min=1, nprobe_max=1, metric=_abi.METRIC_NAMES[metric])
    fl.configure(gemm_variant=variant, grid_workgroups=grid, checksum=True)
    fl.search(q, p, out=out)
    fl.sync()
    never, bad, fsum = fl.census()
    chk = fl.checksum()
    fl.configure(gemm_variant=variant, grid_workgroups=grid, checksum=False, profile=True)
    t0 = time.perf_counter()
    for _ in range(4):
        fl.search(q, p, out=out)
    fl.sync()
    dt = (time.perf_counter() - t0) / 4
    s = fl.stats()
    us = s["us_gemm"] / s["gemm_launches"]
    tf = s["gemm_flops"] / s["gemm_launches"] / us / 1e6
    print(f"variant {s['gemm_variant']} grid {grid} {metric} rows {rows}: gemm {us:.0f} us = {tf:.0f} TF ({tf / 2500:.3f} of peak), "
          f"rest {s['us_rest'] / s['gemm_launches']:.0f} us, step {dt * 1e3:.2f} ms, fallback {s['fallback_queries']}, "
          f"census never={never} bad={bad} sum={fsum:.6e} chk={chk:#x}", flush=True)
