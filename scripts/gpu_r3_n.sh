#!/bin/bash
# round 3, call N: single-query latency A/B of the latency-mode pieces (knob build, one box), then the deferred-refine
# test and the C5 leg again
O=gpurun_out/r3n
mkdir -p $O
export MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_knobs.so
for cfg in "" "MI355_LAT_MERGE_WIDE=0" "MI355_LAT_SLICES_MAX=4" "MI355_LAT_PREFETCH=0" "MI355_LAT_SMALL_FRONT=0" "MI355_LAT_SLICES_MAX=1"; do
  echo "== [$cfg]"; env $cfg timeout 120 python tests/tools/latency_trace.py 2>&1 | grep "single query"
done | tee $O/latency_ab.txt
unset MI355_ANN_LIB
timeout 600 python -m pytest tests/test_gpu_concurrency.py tests/test_gpu_latency_mode.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py --recall-rows 0 --recall2-rows 0 --cpu-seconds 5 --loopback-world 0 > $O/bench_nolegs.json 2> $O/bench_nolegs.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3n/bench_nolegs.json"))
s = d["secondary"]
print("C3", round(d["value"]))
print("latency", json.dumps(s["latency_c3"]))
for k in ("c3_refine10", "c3_refine25", "c5_refine10"):
    print(k, round(s[k]["value"]), s[k]["ms_per_step"], s[k]["stage_us_per_step"], s[k].get("cpu_baseline", {}).get("parity"))
PY
