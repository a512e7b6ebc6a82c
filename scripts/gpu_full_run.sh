#!/bin/bash
# On an MI355X box (gpurun -- 'bash scripts/gpu_full_run.sh [outdir]'): the whole GPU suite, the default bench line end to end
# (wall time printed), rocprofv3 kernel statistics of the C3 and flat commands, smoke(), the single-query tool with the
# planner A/B.  The summaries a round keeps are copied from the out directory into profiles/ by hand.
O=${1:-gpurun_out/full_run}
mkdir -p $O
export PYTHONFAULTHANDLER=1
S=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$? wall=$(( $(date +%s) - S ))s"; tail -4 $O/pytest.txt
S=$(date +%s)
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$? wall=$(( $(date +%s) - S ))s"; tail -3 $O/bench_default.err | cut -c1-300
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c3 -o c3 --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --recall-rows 0 --recall2-rows 0 --secondary 0 --cpu-seconds 0 > $R/$O/prof_c3.log 2>&1
echo "rocprof c3 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_flat -o flat --output-format csv -- python $R/bench.py --workload flat --steps 5 --warmup 1 --cpu-seconds 0 > $R/$O/prof_flat.log 2>&1
echo "rocprof flat rc=$?"
cd $R
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*_kernel_stats.csv"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python - <<'PY'
import glob, json
d = json.load(open(glob.glob("gpurun_out/*/bench_default.json")[-1]))
print("C3", round(d["value"]), "ms", round(d["ms_per_step"], 3), "frac", round(d["roofline"]["frac"], 3))
s = d["secondary"]
print("latency", s["latency_c3"]["single_query_us_eager"], s["latency_c3"]["single_query_stage_us"], s["latency_c3"].get("qps_64_threads_coalesced"))
for k in ("c3_refine10", "c3_refine25", "c5_refine10", "flat_c2_l2", "flat_c2_cosine"):
    if k in s: print(k, round(s[k]["value"]), s[k].get("roofline", {}).get("frac"))
print("loopback", {k: v for k, v in s.get("loopback_world8", {}).get("step_model", {}).items() if k != "note"})
PY
timeout 120 python tests/tools/latency_trace.py 2>&1 | grep "single query"
# planner A/B, if a knob build was shipped along (scripts/build_variants.sh knobs:-DMI355_DEV_KNOBS)
[ -e lancedb_amd/variants/lib_knobs.so ] && bash scripts/ab_variants.sh tests/tools/latency_trace.py knobs knobs:MI355_PLAN_SPARSE=0 | grep "variant\|single query"
