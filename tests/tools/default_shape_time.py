"""Dev: the index shape the reference builds by default — partitions of rows / 8192 (rust/lancedb/src/table/create_index.rs:741-794),
m = dim / 16 (index/vector.rs:306-319), nprobes 20 (query.rs:1103-1104) — on synthetic codes: QPS, stage times, and the A/B of the
batch-level distance tables (csrc/kernels_lut.h) against in-item builds, with row-id checksums.  With a -DMI355_DEV_COUNTERS
build (MI355_ANN_LIB) also the per-item phase split of the scan.
usage: [MI355_ANN_LIB=...] python tests/tools/default_shape_time.py [rows dim m nprobe batch [nlist]]"""
import ctypes as C
import sys

import numpy as np
import torch

import os  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lancedb_amd  # noqa: E402
from lancedb_amd import _abi, _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
m = int(sys.argv[3]) if len(sys.argv) > 3 else dim // 16
nprobes = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "20,64").split(",")]
B = int(sys.argv[5]) if len(sys.argv) > 5 else 2048
nlist = int(sys.argv[6]) if len(sys.argv) > 6 else max(1, n // 8192)
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
q = (centroids[torch.randint(0, nlist, (B,), generator=g, device=dev)] + 0.5 * torch.randn((B, dim), generator=g, device=dev)).contiguous()
torch.cuda.synchronize()
ix = lancedb_amd.IvfPqIndex(centroids, codebook, po, codes, None, codes_layout=_abi.CODES_PART_TRANSPOSED)
del codes
L = _lib.lib()
has_dev = hasattr(L, "mi355_dev_counters")
out = (torch.empty((B, 10), dtype=torch.int64, device=dev), torch.empty((B, 10), dtype=torch.float32, device=dev),
       torch.empty((B,), dtype=torch.int32, device=dev))
print(f"rows {n} dim {dim} m {m} nlist {nlist} batch {B}; partition rows min/median/max {lens.min()}/{int(np.median(lens))}/{lens.max()}", flush=True)
for nprobe in nprobes:
    p = _abi.make_params(k=10, nprobe_min=nprobe, nprobe_max=nprobe)
    for inline in ((False, True, False, True) if not os.environ.get("IMAGES_ONLY") else (False,)):
        ix.configure(profile=0, lut_inline=inline)
        for _ in range(2):
            ix.search(q, p, out=out)
        ix.sync()
        ix.configure(profile=2, lut_inline=inline)
        c = (C.c_uint32 * 8)()
        if has_dev:
            L.mi355_dev_counters(ix._h, c, C.c_int32(1))
        reps = 8
        for _ in range(reps):
            ix.search(q, p, out=out)
        ix.sync()
        st = ix.stats()
        by = st["code_bytes_scanned"] / reps
        line = (f"nprobe {nprobe} {'in-item tables' if inline else 'table images '} (lut_images={st['lut_images']}): "
                f"{B * reps / (st['us_total'] * 1e-6):.0f} QPS | us per launch: coarse {st['us_coarse'] / reps:.0f} select {st['us_select'] / reps:.0f} "
                f"plan+tables {st['us_plan'] / reps:.0f} scan {st['us_scan'] / reps:.0f} merge {st['us_merge'] / reps:.0f} | "
                f"scan {by / (st['us_scan'] / reps) / 1e3:.0f} GB/s of code bytes ({by / (st['us_scan'] / reps) / 1e3 / 8000:.2f} of 8 TB/s; "
                f"with the tables {by / ((st['us_scan'] + st['us_plan']) / reps) / 1e3 / 8000:.2f}) | "
                f"checksum {int(out[0].sum().item())} {float(out[1].double().sum().item()):.6f}")
        if has_dev:
            L.mi355_dev_counters(ix._h, c, C.c_int32(0))
            items = max(c[3], 1)
            line += f" | per item: table {c[0] * 0.01 / items:.2f} us, scan {c[1] * 0.01 / items:.2f} us, merge {c[2] * 0.01 / items:.2f} us ({c[3] // reps} items)"
        print(line, flush=True)
