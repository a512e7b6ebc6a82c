"""Dev: scan time of a 2048-query batch on a synthetic index with SMALL partitions (10 M rows, nlist 4096: ~2.4 k rows each) —
where the per-item distance-table build, not the code stream, decides — with a row-id checksum.
usage: [MI355_ANN_LIB=...] python tests/tools/small_part_time.py [rows nlist batch]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import lancedb_amd  # noqa: E402
from lancedb_amd import _abi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
dim, m, nprobe = 768, 96, 64
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
out = (torch.empty((B, 10), dtype=torch.int64, device=dev), torch.empty((B, 10), dtype=torch.float32, device=dev),
       torch.empty((B,), dtype=torch.int32, device=dev))
p = _abi.make_params(k=10, nprobe_min=nprobe, nprobe_max=nprobe)
ix.configure(profile=0)
for _ in range(2):
    ix.search(q, p, out=out)
ix.sync()
ix.configure(profile=2)
reps = 6
for _ in range(reps):
    ix.search(q, p, out=out)
ix.sync()
st = ix.stats()
print(f"rows {n} nlist {nlist} batch {B}: scan {st['us_scan'] / reps:.0f} us per launch, {B * reps / (st['us_total'] * 1e-6):.0f} QPS, "
      f"checksum {int(out[0].sum().item())} {float(out[1].double().sum().item()):.6f}")
