#!/bin/bash
# round 3, call S: whole suite (weighted plan, merge without dependent loads, cosine/dot row factors), flat cosine A/B,
# single-query latency, loopback leg with warmed ranks
O=gpurun_out/r3s
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?"; grep "passed\|failed" $O/pytest.txt
for lib in base new; do
  [ $lib = new ] && L=$PWD/lancedb_amd/libmi355_ann.so || L=$PWD/lancedb_amd/variants/lib_$lib.so
  echo "== $lib"; MI355_ANN_LIB=$L timeout 300 python tests/tools/flat_gemm_time.py 4000000 4:1:l2 4:1:cosine 4:1:dot 4:256:cosine 2>&1 | grep -v amdgpu.ids
done | tee $O/flat_ab.txt
timeout 120 python tests/tools/latency_trace.py 2>&1 | grep "single query"
timeout 900 python bench.py --recall-rows 0 --recall2-rows 0 --c5-rows 0 --cpu-seconds 0 > $O/bench_loopback.json 2> $O/bench_loopback.err
echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3s/bench_loopback.json"))
lb = d["secondary"]["loopback_world8"]
print("C3", round(d["value"]), "latency", d["secondary"]["latency_c3"]["single_query_us_eager"])
print({k: v for k, v in lb["step_model"].items() if k != "note"})
print("imbalance", lb["overlapped"]["load_imbalance_max_over_mean"], lb["overlapped"]["every_rank_equals_unsharded"])
print([round(p["ms_per_step_wall"], 3) for p in lb["stage_us_per_step_by_rank_alone"]])
PY
