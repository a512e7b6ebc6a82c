#!/bin/bash
# round 3, call L: wide merge / deeper coarse prefetch / slice code prefetch (latency), deferred refine (C5),
# whole suite, the latency + refine + C5 legs, the parity-exposure study, the embedding-like recall set
O=gpurun_out/r3l
mkdir -p $O
export PYTHONFAULTHANDLER=1
S=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$? wall=$(( $(date +%s) - S ))s"; tail -6 $O/pytest.txt
timeout 900 python bench.py --recall-rows 0 --cpu-seconds 5 --loopback-world 0 > $O/bench_nolegs.json 2> $O/bench_nolegs.err
echo "bench rc=$?"; tail -3 $O/bench_nolegs.err | cut -c1-300
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3l/bench_nolegs.json"))
print("C3", round(d["value"]), d["roofline"]["stage_us_per_step"], d["config"].get("timed_region"))
s = d["secondary"]
print("latency", json.dumps(s["latency_c3"]))
for k in ("c3_refine10", "c3_refine25", "c5_refine10"):
    print(k, round(s[k]["value"]), s[k]["stage_us_per_step"], s[k].get("cpu_baseline", {}).get("parity"))
print("recall2", json.dumps(d.get("recall_at_10_embedding_like")))
PY
timeout 600 python tests/tools/parity_exposure.py 2000 > $O/parity_exposure.json 2> $O/parity_exposure.err
echo "parity exposure rc=$?"; tail -2 $O/parity_exposure.err | cut -c1-300; cat $O/parity_exposure.json
