"""Dev: where a single query's wall time goes BETWEEN the kernels.  Reads a rocprofv3 --kernel-trace csv of
tests/tools/latency_trace.py (one search = k_coarse_lat ... the last merge kernel) and prints the medians of every
kernel's duration, of the idle gap in front of it, and of the first-start -> last-end span of a search.

  rocprofv3 --kernel-trace -d OUT -o t --output-format csv -- python tests/tools/latency_trace.py 100000000 4096
  python scripts/trace_gaps.py OUT
"""
import csv
import glob
import os
import sys

import numpy as np

root = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "k_coarse_lat"
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
# searches: runs that start with `first`; "gap": a new search starts behind every idle gap of more than 12 us
searches, cur = [], None
prev_end = None
for s, e, n in rows:
    start = (first in n) if first != "gap" else (prev_end is None or s - prev_end > 12000)
    if start:
        if cur:
            searches.append(cur)
        cur = []
    if cur is not None:
        cur.append((s, e, n))
    prev_end = e
if cur:
    searches.append(cur)
searches = [x for x in searches[len(searches) // 3:] if len(x) == len(searches[2 * len(searches) // 3])][-200:]
print(f"{len(searches)} searches of {len(searches[0])} kernels")
names = [n for _, _, n in searches[0]]
dur = np.array([[e - s for s, e, _ in x] for x in searches]) / 1e3
gap = np.array([[x[i][0] - x[i - 1][1] if i else 0 for i in range(len(x))] for x in searches]) / 1e3
span = np.array([x[-1][1] - x[0][0] for x in searches]) / 1e3
between = np.array([searches[i + 1][0][0] - searches[i][-1][1] for i in range(len(searches) - 1)]) / 1e3
for i, n in enumerate(names):
    print(f"  {n[:60]:60s} gap before {np.median(gap[:, i]):7.2f} us   kernel {np.median(dur[:, i]):7.2f} us")
print(f"  first start -> last end: {np.median(span):.1f} us (kernels {np.median(dur.sum(1)):.1f} + gaps {np.median(gap.sum(1)):.1f});"
      f" last end -> next search's first start: {np.median(between):.1f} us")
