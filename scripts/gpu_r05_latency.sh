#!/bin/bash
# Round-5 latency-path run on an MI355X box: parity tests of the small-batch path, A/B of the new front / table images
# against the knobs that switch them off, and a kernel trace of the single-query path at the C3 shape.
O=${1:-gpurun_out/r05c}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$O
cd $R
timeout 900 python -m pytest tests/test_gpu_latency_mode.py tests/test_gpu_concurrency.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
scripts/ab_variants.sh "tests/tools/latency_trace.py 100000000 4096" knobs knobs:MI355_LAT_FRONT=0 knobs:MI355_LAT_LUT_PRE=0 knobs:MI355_LAT_FRONT=0,MI355_LAT_LUT_PRE=0 knobs:MI355_LAT_SLICES_MAX=16 knobs:MI355_LAT_SLICES_MAX=12 > $O/ab.txt 2>&1
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/lat -o s --output-format csv -- python $R/tests/tools/latency_trace.py 100000000 4096 > $R/$O/lat.log 2>&1; echo "trace rc=$?"
find $R/$O/lat -type f ! -name "*kernel_stats.csv" -delete
grep "single query" $R/$O/lat.log
find $R/$O/lat -name "*kernel_stats.csv" | xargs cat | cut -c1-60,100-200 | head -12
