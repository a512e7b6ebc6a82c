"""Dev tuning harness: one synthetic index, sweep env knobs, print scan-kernel time."""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lancedb_amd
from lancedb_amd import _abi
n, dim, nlist, m, B = int(os.environ.get("N", 25_000_000)), 768, int(os.environ.get("NLIST", 1024)), 96, int(os.environ.get("B", 512))
skew = float(os.environ.get("SKEW", 0.0))
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
cen = torch.randn((nlist, dim), generator=g, device=dev)
cb = torch.randn((m, 256, dim // m), generator=g, device=dev) * 0.5
rng = np.random.default_rng(1)
w = np.exp(rng.normal(0, skew, nlist)); lens = rng.multinomial(n, w / w.sum())
po = np.zeros(nlist + 1, np.uint64); po[1:] = np.cumsum(lens)
codes = torch.randint(0, 256, (n * m,), generator=g, device=dev, dtype=torch.uint8)
ix = lancedb_amd.IvfPqIndex(cen, cb, po, codes, None, codes_layout=_abi.CODES_PART_TRANSPOSED)
del codes
q = (cen[torch.randint(0, nlist, (B,), generator=g, device=dev)] + 0.5 * torch.randn((B, dim), generator=g, device=dev)).contiguous()
params = _abi.make_params(k=10, nprobe_min=64, nprobe_max=64)
print("max_len", lens.max(), "mean", lens.mean())
def run(label, **env):
    for k_, v in env.items(): os.environ[k_] = str(v)
    ix.configure(slice_rows=int(env.get("SLICE", 0)), profile=0)
    ix.search(q, params); torch.cuda.synchronize()
    ix.configure(slice_rows=int(env.get("SLICE", 0)), profile=2)
    t0 = time.perf_counter()
    for _ in range(3): ix.search(q, params)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    st = ix.stats()
    gbs = st["code_bytes_scanned"] / max(st["us_scan"], 1e-9) / 1e3
    print(f"{label:40s} step {dt*1e3:8.2f} ms  scan {st['us_scan']/3/1e3:8.2f} ms  {gbs:7.1f} GB/s  qps {B/dt:8.0f}", flush=True)
    for k_ in env: os.environ.pop(k_, None)
exec(open(sys.argv[1]).read() if len(sys.argv) > 1 else "run('default')")
