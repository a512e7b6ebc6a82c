#!/usr/bin/env python3
"""Dev / CI: compile every translation unit's device code to assembly and list the kernels that use scratch memory
(private_segment_fixed_size > 0).  A register array of HIP `float4` structs with a dozen elements or more is NOT promoted
to registers by hipcc: it lives in scratch, and every global load that fills it is waited for on its own
(`global_load; s_waitcnt vmcnt(0); scratch_store` — round 5 found 24 serial HBM round trips in the latency path's coarse
kernel this way).  Use clang ext-vector types (`__attribute__((ext_vector_type(4))) float`) for such arrays.

    python scripts/check_scratch.py [unit.hip ...]      exit status 1 if a kernel outside the allow-list uses scratch"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# kernels known to keep a little scratch for reasons other than staged loads (checked by hand in the disassembly):
#   k_scan_skew   register spills at the 128-VGPR cap of a 16-wave workgroup; stores in the item prologue, reloads in the
#                 table build and the merge, none in the scan loop (NOTES.md 10.3)
#   k_flat_gemm<  12 bytes in the two-barrier kernel's prologue
ALLOW = ("k_scan_skew", "11k_flat_gemmILi")


def unit_report(src):
    from lancedb_amd import _lib
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "u.s")
        r = subprocess.run([_lib._hipcc()] + _lib.HIPCC_FLAGS + ["-S", "--cuda-device-only", "-o", out, src],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            return src, None, r.stdout[-2000:]
        s = open(out).read()
    rows = []
    # a device function that was NOT inlined (round 6: the scan's selection lambda, called through s_swappc with every captured
    # variable in scratch, after its body had grown past the inliner's threshold — twice the code, and wrong results)
    calls = len(re.findall(r"\bs_swappc_b64\b", s))
    if calls:
        rows.append((f"<{calls} s_swappc_b64 calls: a device function or lambda is not inlined>", 1 << 20, 0))
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, flags=re.S):
        name, body = m.group(1), m.group(2)
        priv = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        vg = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        rows.append((name, priv, vg))
    return src, rows, ""


def main():
    csrc = os.path.join(ROOT, "lancedb_amd", "csrc")
    units = sys.argv[1:] or sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".hip"))
    bad = 0
    with ThreadPoolExecutor(max_workers=4) as ex:
        for src, rows, err in ex.map(unit_report, units):
            if rows is None:
                print("COMPILE FAILED", src, err)
                bad += 1
                continue
            for name, priv, vg in rows:
                if priv:
                    flag = "" if any(a in name for a in ALLOW) else "  <-- scratch"
                    print(f"{os.path.basename(src)}: {name[:90]} scratch {priv} B, {vg} VGPRs{flag}")
                    bad += 0 if any(a in name for a in ALLOW) else 1
    print("kernels with scratch outside the allow-list:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
