#!/bin/bash
O=${1:-gpurun_out/r05d}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$O
cd $R
timeout 600 python -m pytest tests/test_gpu_latency_mode.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
# clocks while single queries run back to back
(MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_knobs.so python -u tests/tools/latency_trace.py 100000000 4096 > $O/bg.log 2>&1 &) ; sleep 45; for i in 1 2 3; do rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -4; sleep 0.05; done > $O/clocks.txt; sleep 8; cat $O/clocks.txt; cat $O/bg.log | grep "single"
MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_dev.so timeout 300 python -u tests/tools/scan_dev_counters.py 100000000 1 2>&1 | grep "^k " > $O/devc_b1.txt; cat $O/devc_b1.txt
MI355_LAT_SLICES_MAX=4 MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_dev.so timeout 300 python -u tests/tools/scan_dev_counters.py 100000000 1 2>&1 | grep "^k 10:" > $O/devc_b1_s4.txt; cat $O/devc_b1_s4.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/lat -o s --output-format csv -- python $R/tests/tools/latency_trace.py 100000000 4096 > $R/$O/lat.log 2>&1; echo "trace rc=$?"
find $R/$O/lat -type f ! -name "*kernel_stats.csv" -delete
grep "single query" $R/$O/lat.log
python - <<PY
import csv
for r in csv.DictReader(open('$R/$O/lat/s_kernel_stats.csv')):
    if r['Name'].startswith(('void k_','k_','__amd')): print(r['Name'][:40], r['Calls'], r['AverageNs'][:8], r['MinNs'], r['MaxNs'])
PY
