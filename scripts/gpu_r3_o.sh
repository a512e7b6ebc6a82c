#!/bin/bash
# round 3, call O: re-rank beside the scan on reserved CUs (C5), latency path without the two pieces that lost their A/B
O=gpurun_out/r3o
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.txt
timeout 120 python tests/tools/latency_trace.py 2>&1 | grep "single query"
timeout 900 python bench.py --recall-rows 0 --recall2-rows 0 --cpu-seconds 5 --loopback-world 0 > $O/bench_nolegs.json 2> $O/bench_nolegs.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3o/bench_nolegs.json"))
s = d["secondary"]
print("C3", round(d["value"]))
print("latency", json.dumps(s["latency_c3"]))
for k in ("c3_refine10", "c3_refine25", "c5_refine10"):
    print(k, round(s[k]["value"]), s[k]["ms_per_step"], s[k]["stage_us_per_step"], s[k].get("cpu_baseline", {}).get("parity"))
PY
