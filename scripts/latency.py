"""Dev: single-query / small-batch latency of the IVF-PQ path at C3 partition sizes
(25 M rows, nlist 1024: same 24 k-row partitions as C3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lancedb_amd
from lancedb_amd import _abi
n, dim, nlist, m = 25_000_000, 768, 1024, 96
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
cen = torch.randn((nlist, dim), generator=g, device=dev)
cb = torch.randn((m, 256, dim // m), generator=g, device=dev) * 0.5
lens = np.random.default_rng(1).multinomial(n, np.ones(nlist) / nlist)
po = np.zeros(nlist + 1, np.uint64); po[1:] = np.cumsum(lens)
codes = torch.randint(0, 256, (n * m,), generator=g, device=dev, dtype=torch.uint8)
ix = lancedb_amd.IvfPqIndex(cen, cb, po, codes, None, codes_layout=_abi.CODES_PART_TRANSPOSED)
del codes
params = _abi.make_params(k=10, nprobe_min=64, nprobe_max=64)
for B in (1, 4, 16, 64, 256):
    q = (cen[torch.randint(0, nlist, (B,), generator=g, device=dev)] + 0.5 * torch.randn((B, dim), generator=g, device=dev)).contiguous()
    hq = q.cpu().numpy()
    for _ in range(3): ix.search(q, params)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): ix.search(q, params); torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20): ix.search(hq, params)
    t_host = (time.perf_counter() - t0) / 20
    print(f"B={B:4d}  device-resident {t_dev*1e6:8.1f} us  host buffers {t_host*1e6:8.1f} us  ({B/t_dev:8.0f} qps)", flush=True)
