"""Dev: A/B of dev knobs on the single-query (host I/O) path, in ONE process over ONE C3-shaped index (the knobs variant of the
library reads its MI355_* environment at every call: scripts/build_variants.sh knobs:-DMI355_DEV_KNOBS, MI355_ANN_LIB).

  python tests/tools/lat_ab.py rows nlist  "" "MI355_DBG_SKIP=4" "MI355_DBG_SKIP=8" ...

Every setting: p50 / p99 of 300 single queries, the batch-of-8 time, and a checksum of the row ids (must not change)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lancedb_amd  # noqa: E402
from lancedb_amd import _abi  # noqa: E402

n = int(sys.argv[1])
nlist = int(sys.argv[2])
settings = sys.argv[3:] or [""]
dim, m = 768, int(os.environ.get("LAT_M", "96"))  # LAT_M=48 LAT_NPROBE=20 with nlist = rows / 8192: the reference's default index shape
nprobe = int(os.environ.get("LAT_NPROBE", "64"))
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
q = (cen[torch.randint(0, nlist, (512,), generator=g, device=dev)] + 0.5 * torch.randn((512, dim), generator=g, device=dev)).cpu().numpy()
kw = dict(k=int(os.environ.get("LAT_K", "10")), nprobe_min=nprobe, nprobe_max=nprobe)
ix.configure(profile=0, graph=False, coalesce=False)
touched = set()
NB = int(os.environ.get("LAT_B", "8"))  # queries of the batch timed beside the single query
for rep in range(2):  # every setting twice, interleaved: the box drifts
    for s in settings:
        for k in touched:
            os.environ.pop(k, None)
        for kv in s.split(","):
            if kv:
                k, v = kv.split("=")
                os.environ[k] = v
                touched.add(k)
        for i in range(10):
            ix.search(q[i:i + 1], **kw)
        lat, chk = [], 0
        for i in range(300):
            t0 = time.perf_counter()
            r = ix.search(q[i:i + 1], **kw)
            lat.append(time.perf_counter() - t0)
            chk ^= int(np.bitwise_xor.reduce(r.rowids.astype(np.uint64).ravel())) * (i + 1) & 0xFFFFFFFFFFFF
        lat = np.sort(np.array(lat)) * 1e6
        t0 = time.perf_counter()
        for i in range(50):
            r8 = ix.search(q[NB * i % 448:NB * i % 448 + NB], **kw)
        b8 = (time.perf_counter() - t0) / 50 * 1e6
        print(f"[{s or 'default':40s}] p50 {lat[150]:6.1f} us  p99 {lat[296]:6.1f} us  batch of {NB}: {b8:6.1f} us  ids {chk:012x}", flush=True)
