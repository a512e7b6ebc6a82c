"""Dev: scan-kernel time of the production ADC scan as a function of kk = k * refine_factor
(candidate-list length), on a C3-shaped index (partitions of ~24 k rows, m = 96, dim 768).
usage: [MI355_ANN_LIB=...] python tests/tools/scan_kk_time.py [rows] [batch]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import lancedb_amd  # noqa: E402
from lancedb_amd import _abi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dim, m, nprobe = 768, 96, 64
nlist = max(64, n // 24_414)
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
ref = {}
for k in (10, 64, 100, 128, 250, 500):
    out = (torch.empty((B, k), dtype=torch.int64, device=dev), torch.empty((B, k), dtype=torch.float32, device=dev),
           torch.empty((B,), dtype=torch.int32, device=dev))
    p = _abi.make_params(k=k, nprobe_min=nprobe, nprobe_max=nprobe)
    ix.configure(profile=0)
    ix.search(q, p, out=out)
    ix.sync()
    ix.configure(profile=2)
    for _ in range(3):
        r = ix.search(q, p, out=out)
    ix.sync()
    st = ix.stats()
    chk = int(r.rowids.to(torch.int64).sum().item()) ^ int(r.distances.double().sum().item() * 1000)
    print(f"k {k}: scan {st['us_scan'] / 3:.0f} us, merge {st['us_merge'] / 3:.0f} us per {B}-query launch, result checksum {chk}", flush=True)
