"""Dev: 200 single-query flat searches on a 100 k x 128 f32 table (BASELINE configs[0]) — run under
rocprofv3 --kernel-trace --stats to see which kernels the call is made of.  usage: python tests/tools/flat_small_trace.py [path]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import lancedb_amd  # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "auto" else None
rng = np.random.default_rng(3)
v = rng.random((100_000, 128), dtype=np.float32)
qs = rng.random((64, 128), dtype=np.float32)
fl = lancedb_amd.FlatIndex(v)
fl.configure(path=path)
for i in range(5):
    fl.search(qs[i:i + 1], k=10)
t0 = time.perf_counter()
for i in range(200):
    fl.search(qs[i % 64:i % 64 + 1], k=10)
print(f"path {path or 'auto'}: {(time.perf_counter() - t0) / 200 * 1e6:.0f} us per host-I/O single query (took {fl.info()[0]})")
