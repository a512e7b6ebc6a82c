"""Dev: the scan's per-item phase times and selection counters (a -DMI355_DEV_COUNTERS build) on the TRAINED index of the
bench's recall leg (bench.recall_index: Gaussian-mixture column, k-means IVF + residual PQ) next to a synthetic index of the
same size — real partitions and real distance distributions admit rows the uniform-random codes never do.
usage: MI355_ANN_LIB=lancedb_amd/variants/lib_dev.so python tests/tools/trained_dev_counters.py [rows]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
import lancedb_amd  # noqa: E402
from lancedb_amd import _abi, _lib  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sys.argv = [sys.argv[0], "--recall-rows", str(rows), "--recall-queries", "4096"]
a = bench.parse()
R = bench.recall_index(a, 768, 96)
dev = R["x"].device
ix = lancedb_amd.IvfPqIndex(R["cen"].contiguous(), R["codebook"].contiguous(), R["part_offsets"], R["codes"], R["order"])
po = np.asarray(R["part_offsets"]).astype(np.int64)
lens = np.diff(po)
print(f"partitions: {len(lens)}, rows min / median / mean / max = {lens.min()} / {int(np.median(lens))} / {lens.mean():.0f} / {lens.max()}, empty {int((lens == 0).sum())}")
q = R["q"][:2048].contiguous()
L = _lib.lib()
has_dev = hasattr(L, "mi355_dev_counters")
out = (torch.empty((2048, 10), dtype=torch.int64, device=dev), torch.empty((2048, 10), dtype=torch.float32, device=dev),
       torch.empty((2048,), dtype=torch.int32, device=dev))
p = _abi.make_params(k=10, nprobe_min=64, nprobe_max=64)
ix.configure(profile=0)
ix.search(q, p, out=out)
ix.sync()
ix.configure(profile=2)
c = (C.c_uint32 * 8)()
if has_dev:
    L.mi355_dev_counters(ix._h, c, C.c_int32(1))
reps = 3
for _ in range(reps):
    ix.search(q, p, out=out)
ix.sync()
st = ix.stats()
line = f"trained {rows}: scan {st['us_scan'] / reps:.0f} us, coarse {st['us_coarse'] / reps:.0f}, select {st['us_select'] / reps:.0f}, merge {st['us_merge'] / reps:.0f} us per 2048-query launch; rows scanned per query {st['vectors_scanned'] / reps / 2048:.0f}"
if has_dev:
    L.mi355_dev_counters(ix._h, c, C.c_int32(0))
    items = max(c[3], 1)
    t = 0.01
    line += (f" | per item: lut {c[0] * t / items:.1f} us, scan {c[1] * t / items:.1f} us, merge {c[2] * t / items:.1f} us; items {c[3] // reps}, "
             f"in lists at merge/item {c[4] / items:.1f}, barrier->ranking {c[6] * t / items:.1f} us, ranking {c[7] * t / items:.1f} us")
print(line, flush=True)
