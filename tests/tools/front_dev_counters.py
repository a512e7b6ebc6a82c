"""Dev: stage times INSIDE the latency front's two kernels (k_coarse_lat block samples, k_select_plan) for single queries on a
C3-shaped index, from a -DMI355_DEV_FRONT build (wall_clock64 stamps, 10 ns).
usage: MI355_ANN_LIB=lancedb_amd/variants/lib_front.so python tests/tools/front_dev_counters.py [rows nlist]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import lancedb_amd  # noqa: E402
from lancedb_amd import _abi, _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # queries per call (the planning workgroup of a batch is the LAST one to finish its selection)
dim, m = 768, 96
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
kw = dict(k=10, nprobe_min=64, nprobe_max=64)
ix.configure(profile=0, graph=False, coalesce=False)
L = _lib.lib()
for i in range(10):
    ix.search(q[i:i + B], **kw)
c = (C.c_uint32 * 8)()
L.mi355_dev_counters(ix._h, c, C.c_int32(1))
N = 200
for i in range(N):
    ix.search(q[i:i + B], **kw)
L.mi355_dev_counters(ix._h, c, C.c_int32(0))
t = 0.01
s = max(c[3], 1)
print(f"k_coarse_lat, per sampled block ({c[3]} samples): query->LDS {c[0] * t / s:.2f} us, row loads + staging {c[1] * t / s:.2f} us, chains {c[2] * t / s:.2f} us")
print(f"k_select_plan, per call: keys {c[4] * t / N:.2f} us, radix windows {c[5] * t / N:.2f} us, emit + ticket {c[6] * t / N:.2f} us, plan {c[7] * t / N:.2f} us")
print(f"(a -DMI355_DEV_PLAN build instead: plan_sparse_body per call: loads + totals {c[0] * t / N:.2f} us, T + histogram {c[1] * t / N:.2f} us, rank loop {c[2] * t / N:.2f} us, emit {c[3] * t / N:.2f} us)")
