"""Dev: where a scan work item's time goes as a function of kk = k * refine_factor, from the counters a
-DMI355_DEV_COUNTERS build keeps (kernels_skew.h SK_DEV): per-item LUT / scan / merge time, rows admitted to the
candidate lists, rows in the lists at the merge, optimistic passes redone, items that ran without a query bound.
usage: MI355_ANN_LIB=lancedb_amd/variants/lib_dev.so python tests/tools/scan_dev_counters.py [rows] [batch] [nlist]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import lancedb_amd  # noqa: E402
from lancedb_amd import _abi, _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dim, m, nprobe = 768, 96, 64
nlist = int(sys.argv[3]) if len(sys.argv) > 3 else max(64, n // 24_414)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(0x1A2CE)
centroids = torch.randn((nlist, dim), generator=g, device=dev)
codebook = torch.randn((m, 256, dim // m), generator=g, device=dev) * 0.5
rng = np.random.default_rng(0x1A2CE)
w = np.exp(rng.normal(0.0, 0.5, size=nlist))
lens = rng.multinomial(n, w / w.sum())
po = np.zeros(nlist + 1, dtype=np.uint64)
po[1:] = np.cumsum(lens)
codes = torch.randint(0, 256, (n * m,), generator=g, device=dev, dtype=torch.uint8)
rid = torch.randperm(n, generator=g, device=dev)
q = (centroids[torch.randint(0, nlist, (B,), generator=g, device=dev)] + 0.5 * torch.randn((B, dim), generator=g, device=dev)).contiguous()
torch.cuda.synchronize()
ix = lancedb_amd.IvfPqIndex(centroids, codebook, po, codes, rid, codes_layout=_abi.CODES_PART_TRANSPOSED)
del codes
L = _lib.lib()
has_dev = hasattr(L, "mi355_dev_counters")
for k in (10, 100, 250, 500):
    out = (torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
           torch.empty((B,), dtype=torch.int32, device=dev))
    p = _abi.make_params(k=k, nprobe_min=nprobe, nprobe_max=nprobe)
    ix.configure(profile=0)
    ix.search(q, p, out=out)
    ix.sync()
    ix.configure(profile=2)
    c = (C.c_uint32 * 8)()
    if has_dev:
        L.mi355_dev_counters(ix._h, c, C.c_int32(1))
    reps = 3
    for _ in range(reps):
        r = ix.search(q, p, out=out)
    ix.sync()
    st = ix.stats()
    line = f"k {k}: scan {st['us_scan'] / reps:.0f} us, merge {st['us_merge'] / reps:.0f} us per {B}-query launch"
    if has_dev:
        L.mi355_dev_counters(ix._h, c, C.c_int32(0))
        items = max(c[3], 1)
        tick_us = 0.01  # wall_clock64: 100 MHz
        line += (f" | per item: lut {c[0] * tick_us / items:.1f} us, scan {c[1] * tick_us / items:.1f} us, merge {c[2] * tick_us / items:.1f} us; "
                 f"items {c[3] // reps}, in lists at merge/item {c[4] / items:.0f}, "
                 f"{'redone passes ' + str(c[5] // reps) if k > 128 else 'shader clock over the items ' + format(c[5] / max(c[0] + c[1] + c[2], 1) * 100, '.0f') + ' MHz'}, "
                 f"merge split: barrier->ranking {c[6] * tick_us / items:.1f} us, ranking {c[7] * tick_us / items:.1f} us (thread 0)")
        import os
        if os.environ.get("SCANSPLIT"):
            line += (f" || SCANSPLIT build, wave 0 per item: first chunk {c[4] * tick_us / items:.2f} us, positions {c[6] * tick_us / items:.2f} us, "
                     f"tail {c[7] * tick_us / items:.2f} us (the three numbers after 'in lists' / 'merge split' are these)")
    print(line, flush=True)
