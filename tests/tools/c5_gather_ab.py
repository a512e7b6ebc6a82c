"""Dev: the C5 leg alone (bench.c5_refine10) at reduced rows, for A/Bs of the PCIe re-rank: `--c5-hugepages 1`, and with a
-DMI355_DEV_KNOBS library MI355_REFINE_GATHER=0|1 (one row per lane / eight lanes per 128 B of a row).
usage: python tests/tools/c5_gather_ab.py [bench.py arguments, e.g. --c5-rows 20000000 --c5-hugepages 1]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

sys.argv = [sys.argv[0]] + sys.argv[1:]
a = bench.parse()
a.cpu_seconds = 0
torch.cuda.set_device(0)
r = bench.c5_refine10(a, torch, np, torch.device("cuda", 0))
print(json.dumps({"qps": r.get("value"), "stage_us": r.get("stage_us_per_step"), "gather": r.get("refine_gather"),
                  "rows": r.get("config", {}).get("n_rows"), "hugepages": a.c5_hugepages, "column": a.c5_column,
                  "page_lock_s": r.get("config", {}).get("raw_page_lock_s"), "fill_s": r.get("config", {}).get("raw_fill_s")}))
