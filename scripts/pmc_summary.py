"""Dev: per-launch averages of the PMC counters rocprofv3 collected for one kernel.
usage: python scripts/pmc_summary.py <dir with <pass>/p_counter_collection.csv> <kernel substring>"""
import collections
import csv
import glob
import os
import sys

csv.field_size_limit(1 << 30)
root, needle = sys.argv[1], sys.argv[2]
for path in sorted(glob.glob(os.path.join(root, "*", "p_counter_collection.csv"))):
    acc, n = collections.defaultdict(float), collections.defaultdict(int)
    dur = []
    for r in csv.DictReader(open(path)):
        if needle not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        n[r["Counter_Name"]] += 1
        dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    name = os.path.basename(os.path.dirname(path))
    if not acc:
        print(name, "no rows for", needle)
        continue
    print(name, "launches", max(n.values()), "avg_us", round(sum(dur) / len(dur) / 1e3, 1),
          {k: round(v / n[k], 1) for k, v in sorted(acc.items())})
