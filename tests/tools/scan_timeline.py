"""Dev: the per-workgroup timeline of the LAST scan launch of a single query on a C3-shaped index (a -DMI355_DEV_TIMELINE build:
scripts/build_variants.sh tl:-DMI355_DEV_KNOBS,-DMI355_DEV_TIMELINE; MI355_ANN_LIB=lancedb_amd/variants/lib_tl.so).
Prints where the launch's wall time goes: per item the table / scan (own wave, slowest wave) / merge phases, the spread of the
workgroups' start and end times, and what the workgroups of each XCD did.
usage: python tests/tools/scan_timeline.py [rows nlist batch] [KNOB=v,...]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lancedb_amd  # noqa: E402
from lancedb_amd import _abi, _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
settings = sys.argv[4:] or [""]
dim, m = 768, int(os.environ.get("LAT_M", "96"))  # (LAT_M / LAT_NPROBE / LAT_K: like lat_ab.py)
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
nprobe = int(os.environ.get("LAT_NPROBE", "64"))
kw = dict(k=int(os.environ.get("LAT_K", "10")), nprobe_min=nprobe, nprobe_max=nprobe)
ix.configure(profile=0, graph=False, coalesce=False)
L = _lib.lib()
L.mi355_dev_timeline.restype = C.c_int32
TICK_US = 1.0 / 100.0  # wall_clock64: 100 MHz
touched = set()
for s in settings:
    for k in touched:
        os.environ.pop(k, None)
    for kv in s.split(","):
        if kv:
            k, v = kv.split("=")
            os.environ[k] = v
            touched.add(k)
    print(f"=== [{s or 'default'}] batch {B}")
    acc = []
    for rep in range(12):
        ix.search(q[rep * B:(rep + 1) * B], **kw)
        buf = (C.c_uint64 * (1024 * 64))()
        grid, words = C.c_uint32(0), C.c_uint32(0)
        assert L.mi355_dev_timeline(ix._h, buf, C.c_uint32(1024 * 64), C.byref(grid), C.byref(words)) == 0
        t = np.frombuffer(buf, dtype=np.uint64)[: grid.value * words.value].reshape(grid.value, words.value).astype(np.int64)
        if rep < 4:
            continue  # warm-up
        t0 = t[:, 0].min()
        n_items = (t[:, 1] >> 8) & 0xFF
        xcc = t[:, 1] & 0xFF
        it = t[:, 2:].reshape(grid.value, -1, 6)
        wg_end = np.array([it[g_, n_items[g_] - 1, 4] if n_items[g_] else t[g_, 0] for g_ in range(grid.value)])
        span = (wg_end.max() - t0) * TICK_US
        ph = {"table": [], "scan own wave": [], "wait slowest wave": [], "merge": [], "item": [], "KB": []}
        for g_ in range(grid.value):
            for i in range(min(n_items[g_], it.shape[1])):
                a0, a1, a2, a3, a4, meta = it[g_, i]
                ph["table"].append((a1 - a0) * TICK_US)
                ph["scan own wave"].append((a2 - a1) * TICK_US)
                ph["wait slowest wave"].append((a3 - a2) * TICK_US)
                ph["merge"].append((a4 - a3) * TICK_US)
                ph["item"].append((a4 - a0) * TICK_US)
                ph["KB"].append((meta >> 32) * m / 1024.0 / ((((meta & 0xFFFFFFFF) >> 26) & 63) + 1 if B * 64 < 768 else 1))
        acc.append((span, (t[:, 0].max() - t0) * TICK_US, (wg_end.min() - t0) * TICK_US, np.mean(n_items), {k: np.array(v) for k, v in ph.items()}, n_items, xcc, wg_end, t0, t, it))
    span = np.median([a[0] for a in acc])
    print(f"launch span (first workgroup start -> last workgroup end): {span:.1f} us; last workgroup STARTS at {np.median([a[1] for a in acc]):.1f} us;"
          f" first workgroup is out of work at {np.median([a[2] for a in acc]):.1f} us; items per workgroup {acc[-1][3]:.2f}")
    for k in ("table", "scan own wave", "wait slowest wave", "merge", "item", "KB"):
        v = np.concatenate([a[4][k] for a in acc])
        print(f"  {k:18s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f}")
    # the last repetition in detail: per XCD
    span_, _, _, _, ph, n_items, xcc, wg_end, t0, t, it = acc[-1]
    for x in range(8):
        sel = xcc == x
        if not sel.any():
            continue
        print(f"  XCD {x}: {sel.sum():3d} workgroups, items {n_items[sel].sum():3d}, busy until {((wg_end[sel] - t0) * TICK_US).min():6.1f} .. {((wg_end[sel] - t0) * TICK_US).max():6.1f} us")
    order = np.argsort(wg_end)
    print("  the five workgroups that finish last (item: start, table, scan, merge end; KB):")
    for g_ in order[-5:]:
        line = f"    wg {g_:3d} xcd {xcc[g_]}:"
        for i in range(n_items[g_]):
            a0, a1, a2, a3, a4, meta = it[g_, i]
            line += f"  [{(a0 - t0) * TICK_US:5.1f} {(a1 - t0) * TICK_US:5.1f} {(a3 - t0) * TICK_US:5.1f} {(a4 - t0) * TICK_US:5.1f}; {(meta >> 32) * m / 1024.0 / ((((meta & 0xFFFFFFFF) >> 26) & 63) + 1 if B * 64 < 768 else 1):5.0f} KB]"
        print(line)
