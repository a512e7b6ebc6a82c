"""Dev: the two exact flat paths (bf16 MFMA filter + re-rank / exact sweep) timed per call over table sizes and batch
sizes, device I/O — the numbers behind the cost model of run_flat_search_device (csrc/ann_flat.hip).
usage: python tests/tools/flat_path_time.py [rows:dim:dtype ...]   (dtype f32 | bf16)"""
import sys
import time

import torch

sys.path.insert(0, ".")
import lancedb_amd  # noqa: E402
from lancedb_amd import _abi  # noqa: E402

shapes = sys.argv[1:] or ["100000:128:f32", "1000000:128:f32", "1000000:768:bf16", "10000000:768:bf16"]
dev = torch.device("cuda", 0)
for spec in shapes:
    n, dim, dt = spec.split(":")
    n, dim = int(n), int(dim)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    if dt == "bf16":
        col = torch.randn((n, dim), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
        fl = lancedb_amd.FlatIndex(col.view(torch.int16), dtype=_abi.DTYPE_BF16, device=0)
    else:
        col = torch.randn((n, dim), generator=g, device=dev, dtype=torch.float32)
        fl = lancedb_amd.FlatIndex(col)
    fl.set_stream(torch.cuda.current_stream().cuda_stream)
    for nq in (1, 2, 4, 8, 16, 64, 256):
        q = torch.randn((nq, dim), generator=g, device=dev)
        out = (torch.empty((nq, 10), dtype=torch.int64, device=dev), torch.empty((nq, 10), dtype=torch.float32, device=dev),
               torch.empty((nq,), dtype=torch.int32, device=dev))
        line, sums = [], []
        for path in ("filter", "sweep", None):
            fl.configure(path=path)
            reps = 3 if (path == "sweep" and nq * n * dim > 2e10) else 10
            if path == "sweep" and nq * n * dim > 3e11:
                line.append("sweep      -")
                sums.append(None)
                continue
            for _ in range(2):
                fl.search(q, k=10, out=out)
            fl.sync()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fl.search(q, k=10, out=out)
            fl.sync()
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / reps * 1e6
            sums.append(int(out[0].sum().item()))
            line.append(f"{path or 'auto'} {us:9.0f} us" + (f" (took {'filter' if fl.info()[0] == 1 else 'sweep'})" if path is None else ""))
        same = len({x for x in sums if x is not None}) == 1
        print(f"{spec} nq {nq:4d}: " + "   ".join(line) + ("" if same else "   RESULTS DIFFER"), flush=True)
    fl.close()
    del col
    torch.cuda.empty_cache()
