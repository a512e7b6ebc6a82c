"""Dev: what HBM write bandwidth a plain fill reaches on this box (the ceiling of kernels that only write, e.g. the
batch-level distance-table kernel's 2 GB of images per 2048 x 20-pair batch).  usage: python tests/tools/write_bw_probe.py"""
import time

import torch

dev = torch.device("cuda", 0)
for gb in (0.5, 2, 8):
    n = int(gb * (1 << 30)) // 4
    x = torch.empty(n, dtype=torch.float32, device=dev)
    y = torch.empty(n, dtype=torch.float32, device=dev)
    for name, fn in (("fill", lambda: x.fill_(1.0)), ("copy", lambda: y.copy_(x))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        moved = n * 4 * (2 if name == "copy" else 1)
        print(f"{name} {gb} GB: {dt * 1e3:.3f} ms, {moved / dt / 1e12:.2f} TB/s ({'read + write' if name == 'copy' else 'write only'})", flush=True)
    del x, y
