"""Dev: single-query (host I/O) searches on C3-shaped partitions, for a rocprofv3 --kernel-trace --stats run:
per-kernel device times of the latency path next to the wall-clock p50 / p99."""
import sys
import time

import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lancedb_amd  # noqa: E402
from lancedb_amd import _abi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000  # 100000000 4096: the C3 shape
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # queries per call
dim, m = 768, int(os.environ.get("LAT_M", "96"))  # LAT_M=48 LAT_NPROBE=20, nlist = rows / 8192: the reference's default shape
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
for c0 in range(0, n * m, 1 << 30):  # in pieces: one randint call over 9.6e9 elements (the C3 shape) left the GPU faulting
    c1 = min(n * m, c0 + (1 << 30))
    torch.randint(0, 256, (c1 - c0,), generator=g, device=dev, dtype=torch.uint8, out=codes[c0:c1])
torch.cuda.synchronize()
ix = lancedb_amd.IvfPqIndex(cen, cb, po, codes, None, codes_layout=_abi.CODES_PART_TRANSPOSED)
del codes
q = (cen[torch.randint(0, nlist, (512,), generator=g, device=dev)] + 0.5 * torch.randn((512, dim), generator=g, device=dev)).cpu().numpy()
nprobe = int(os.environ.get("LAT_NPROBE", "64"))
kw = dict(k=10, nprobe_min=nprobe, nprobe_max=nprobe)
ix.configure(profile=0, graph=False, coalesce=False)
for i in range(10):
    ix.search(q[i:i + B], **kw)
lat = []
for i in range(300):
    t0 = time.perf_counter()
    ix.search(q[i:i + B], **kw)
    lat.append(time.perf_counter() - t0)
lat = np.sort(np.array(lat)) * 1e6
print(f"{B} queries per call, host I/O: p50 {lat[150]:.1f} us  p99 {lat[296]:.1f} us  mean {lat.mean():.1f} us", flush=True)
