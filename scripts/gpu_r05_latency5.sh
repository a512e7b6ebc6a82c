#!/bin/bash
O=${1:-gpurun_out/r05j}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$O
cd $R
timeout 600 python -m pytest tests/test_gpu_latency_mode.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log | cut -c1-300
MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_front.so timeout 250 python -u tests/tools/front_dev_counters.py 100000000 4096 2>&1 | grep "^k_" > $O/front.txt; cat $O/front.txt
scripts/ab_variants.sh "tests/tools/latency_trace.py 100000000 4096" knobs lut24 knobs:MI355_LAT_FRONT=0 > $O/ab.txt 2>&1; cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/lat -o s --output-format csv -- python $R/tests/tools/latency_trace.py 100000000 4096 > $R/$O/lat.log 2>&1; echo "trace rc=$?"
find $R/$O/lat -type f ! -name "*kernel_stats.csv" -delete
grep "single query" $R/$O/lat.log
python - <<PY
import csv
for r in csv.DictReader(open('$R/$O/lat/s_kernel_stats.csv')):
    if r['Name'].startswith(('void k_','k_','__amd')): print(r['Name'][:40], r['Calls'], r['AverageNs'][:8], r['MinNs'], r['MaxNs'])
PY
