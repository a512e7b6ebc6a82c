"""Dev: which path the block merge of a single query takes (dev[5] of a -DMI355_DEV_COUNTERS build: 7e8 + filled slots = one row per
thread; short rows + 1e5 * final rows = sweep + selection; + 1e9 = one-wave fallback) and the scan's per-item phase ticks.
usage: MI355_ANN_LIB=lancedb_amd/variants/lib_dev.so LAT_M=48 LAT_NPROBE=20 python tests/tools/merge_dev_counters.py rows nlist"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lancedb_amd  # noqa: E402
from lancedb_amd import _abi, _lib  # noqa: E402

n, nlist = int(sys.argv[1]), int(sys.argv[2])
dim, m, nprobe = 768, int(os.environ.get("LAT_M", "96")), int(os.environ.get("LAT_NPROBE", "64"))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(1)
cen = torch.randn((nlist, dim), generator=g, device=dev)
cb = torch.randn((m, 256, dim // m), generator=g, device=dev) * 0.5
rng = np.random.default_rng(1)
w = np.exp(rng.normal(0.0, 0.5, size=nlist))
lens = rng.multinomial(n, w / w.sum())
po = np.zeros(nlist + 1, np.uint64)
po[1:] = np.cumsum(lens)
codes = torch.empty((n * m,), device=dev, dtype=torch.uint8)
for c0 in range(0, n * m, 1 << 30):
    c1 = min(n * m, c0 + (1 << 30))
    torch.randint(0, 256, (c1 - c0,), generator=g, device=dev, dtype=torch.uint8, out=codes[c0:c1])
torch.cuda.synchronize()
ix = lancedb_amd.IvfPqIndex(cen, cb, po, codes, None, codes_layout=_abi.CODES_PART_TRANSPOSED)
del codes
q = (cen[torch.randint(0, nlist, (64,), generator=g, device=dev)] + 0.5 * torch.randn((64, dim), generator=g, device=dev)).cpu().numpy()
ix.configure(profile=0, graph=False, coalesce=False)
L = _lib.lib()
for i in range(5):
    ix.search(q[i:i + 1], k=10, nprobe_min=nprobe, nprobe_max=nprobe)
c = (C.c_uint32 * 8)()
L.mi355_dev_counters(ix._h, c, C.c_int32(1))
for i in range(8):
    ix.search(q[8 + i:9 + i], k=10, nprobe_min=nprobe, nprobe_max=nprobe)
    L.mi355_dev_counters(ix._h, c, C.c_int32(1))
    v = list(c)
    print(f"query {i}: items {v[3]}, per item table {v[0] / max(v[3], 1) / 100:.1f} us scan {v[1] / max(v[3], 1) / 100:.1f} us merge {v[2] / max(v[3], 1) / 100:.1f} us; "
          f"rows in lists at the merge {v[4]}, dev[5] (merge path) {v[5]}, dev[6] {v[6]}, dev[7] {v[7]}")
