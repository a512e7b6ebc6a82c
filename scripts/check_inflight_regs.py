#!/usr/bin/env python3
"""Static check of the SLABBED k_scan_skew instantiations (csrc/kernels_skew.h; table-building and table-image kernels): the partial-sum register px is loaded
by inline asm (global_load_dwordx2) and consumed a tile later behind a counted s_waitcnt.  hipcc does not know the
register is in flight; if its register allocator ever copies px (v_mov) between the load and the wait, the copy reads
stale data.  This script compiles the translation unit to assembly and fails if any such copy exists.

    python scripts/check_inflight_regs.py        (run by tests/test_build_host.py)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from lancedb_amd import _lib
    # both families of SLABBED kernels: the ones that build their distance tables and the ones that copy table images
    s = ""
    procs = []
    with tempfile.TemporaryDirectory() as td:
        for unit in ("ann_scan_skew_slab.hip", "ann_scan_skew_slab_img.hip"):
            src = os.path.join(ROOT, "lancedb_amd", "csrc", unit)
            out = os.path.join(td, unit + ".s")
            cmd = [_lib._hipcc()] + _lib.HIPCC_FLAGS + ["-S", "--cuda-device-only", "-o", out, src]
            procs.append((out, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        for out, pr in procs:
            log, _ = pr.communicate()
            if pr.returncode != 0:
                print(log[-3000:])
                return 2
            s += open(out).read()
    bad, n_kernels, n_loads = [], 0, 0
    for name in re.findall(r"\n(_Z11k_scan_skew\w+):", s):
        i = s.index("\n" + name + ":")
        f = s[i:s.index(".Lfunc_end", i)].split("\n")
        n_kernels += 1
        for n, line in enumerate(f):
            m = re.match(r"\s*global_load_dwordx2 (v\[\d+:\d+\]), v\[\d+:\d+\], off\s*$", line)
            if not m:
                continue
            n_loads += 1
            reg = m.group(1)
            lo, hi = (int(x) for x in reg[2:-1].split(":"))
            names = {reg, f"v{lo}", f"v{hi}"}
            # walk forward (straight-line and through labels) until a wait that can retire the load
            for k in range(n + 1, min(n + 400, len(f))):
                t = f[k].strip()
                if t.startswith("s_waitcnt") and "vmcnt" in t:
                    break
                if t.startswith("s_branch") or t.startswith("s_cbranch") or t.startswith("s_endpgm"):
                    break  # (the wait opens the next tile: the walk ends at the loop's back edge)
                mm = re.match(r"v_mov_b(32|64)_e32 (\S+), (\S+)", t)
                if mm and (mm.group(3) in names or mm.group(2).rstrip(",") in names):
                    bad.append((name, n, k, t))
    print(f"{n_kernels} kernels, {n_loads} in-flight register loads, {len(bad)} copies of a register in flight")
    for b in bad[:10]:
        print("  ", b)
    return 1 if bad or not n_loads else 0


if __name__ == "__main__":
    sys.exit(main())
