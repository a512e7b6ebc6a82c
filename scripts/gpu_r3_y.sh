#!/bin/bash
# round 3, call Y: block merge + fused planner: suite, single-query latency with A/B knobs, kernel stats
O=gpurun_out/r3y
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$?"; grep "passed\|failed" $O/pytest.txt; tail -5 $O/pytest.txt
timeout 120 python tests/tools/latency_trace.py 2>&1 | grep "single query"
for kn in "" "MI355_MERGE_BLOCK_MAX_NQ=0"; do
  echo "== knobs [$kn]"; env $kn MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_knobs.so timeout 120 python tests/tools/latency_trace.py 2>&1 | grep "single query"
done
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/lat_prof -o lat --output-format csv -- python $R/tests/tools/latency_trace.py > $R/$O/lat_prof.log 2>&1)
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r3y/lat_prof/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
echo "== dev counters, one query"
MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_dev.so timeout 200 python tests/tools/scan_dev_counters.py 25000000 1 2>&1 | grep "^k "
