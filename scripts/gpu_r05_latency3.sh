#!/bin/bash
O=${1:-gpurun_out/r05f}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$O
cd $R
timeout 600 python -m pytest tests/test_gpu_latency_mode.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_dev.so timeout 300 python -u tests/tools/scan_dev_counters.py 100000000 1 2>&1 | grep "^k 10:" > $O/devc_b1.txt; cat $O/devc_b1.txt
C3="python bench.py --steps 10 --warmup 2 --recall-rows 0 --recall2-rows 0 --secondary 0 --cpu-seconds 0"
for v in "" lut16 lut24; do
  if [ -z "$v" ]; then lib=""; else lib="MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_$v.so"; fi
  echo "== c3 [$v]"; env $lib $C3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['us_per_launch'], d['roofline']['frac'])"
done > $O/c3_lut_inflight.txt 2>&1
cat $O/c3_lut_inflight.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/lat -o s --output-format csv -- python $R/tests/tools/latency_trace.py 100000000 4096 > $R/$O/lat.log 2>&1; echo "trace rc=$?"
find $R/$O/lat -type f ! -name "*kernel_stats.csv" -delete
grep "single query" $R/$O/lat.log
python - <<PY
import csv
for r in csv.DictReader(open('$R/$O/lat/s_kernel_stats.csv')):
    if r['Name'].startswith(('void k_','k_','__amd')): print(r['Name'][:40], r['Calls'], r['AverageNs'][:8], r['MinNs'], r['MaxNs'])
PY
cd $R; python -u tests/tools/latency_trace.py 100000000 4096 2>&1 | grep "single"
