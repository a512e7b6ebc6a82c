#!/bin/bash
# round 3, call K: latency path (fused prep + coarse, parallel probe select, grouped merge loads, 8 slices), fast
# distance-table build for 16-float sub-vectors (C5); suite subsets + the latency / C5 legs of the bench
O=gpurun_out/r3k
mkdir -p $O
export PYTHONFAULTHANDLER=1
S=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$? wall=$(( $(date +%s) - S ))s"; tail -6 $O/pytest.txt
timeout 900 python bench.py --recall-rows 0 --recall2-rows 0 --cpu-seconds 5 --loopback-world 0 > $O/bench_nolegs.json 2> $O/bench_nolegs.err
echo "bench rc=$?"; tail -3 $O/bench_nolegs.err | cut -c1-300
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3k/bench_nolegs.json"))
print("C3", round(d["value"]), d["roofline"]["stage_us_per_step"])
s = d["secondary"]
print("latency", json.dumps(s["latency_c3"]))
for k in ("c3_refine10", "c3_refine25", "c5_refine10"):
    print(k, round(s[k]["value"]), s[k]["stage_us_per_step"], s[k].get("cpu_baseline", {}).get("parity"))
PY
