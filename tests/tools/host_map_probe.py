"""Dev probe for the C5 line: how long does it take to fault in, fill and page-lock (hipHostRegister,
mapped) a large host buffer on this box, and what does a random 3-KiB-row gather over PCIe deliver?"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lancedb_amd import _hip  # noqa: E402

rt = _hip.runtime()
GB = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = GB << 30
t0 = time.perf_counter()
buf = np.empty(n, dtype=np.uint8)
blk = np.random.default_rng(0).integers(0, 255, size=1 << 28, dtype=np.uint8)
t1 = time.perf_counter()
import threading
def fill(lo, hi):
    for o in range(lo, hi, blk.size):
        e = min(hi, o + blk.size)
        buf[o:e] = blk[:e - o]
T = 32
th = [threading.Thread(target=fill, args=(n * i // T, n * (i + 1) // T)) for i in range(T)]
[t.start() for t in th]; [t.join() for t in th]
t2 = time.perf_counter()
print(f"alloc+block {t1 - t0:.2f} s, fill {GB} GB with {T} threads {t2 - t1:.2f} s = {GB / (t2 - t1):.1f} GB/s")
rt.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
t3 = time.perf_counter()
rc = rt.hipHostRegister(buf.ctypes.data_as(C.c_void_p), C.c_size_t(n), C.c_uint(2))  # hipHostRegisterMapped
t4 = time.perf_counter()
print(f"hipHostRegister({GB} GB) rc={rc} {t4 - t3:.2f} s = {GB / (t4 - t3):.1f} GB/s")
if rc == 0:
    t5 = time.perf_counter()
    rt.hipHostUnregister(buf.ctypes.data_as(C.c_void_p))
    print(f"unregister {time.perf_counter() - t5:.2f} s")
