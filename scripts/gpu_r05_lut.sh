#!/bin/bash
O=${1:-gpurun_out/r05k}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$O
cd $R
C3="python bench.py --steps 10 --warmup 2 --recall-rows 0 --recall2-rows 0 --secondary 0 --cpu-seconds 0"
for v in knobs lut12 lut16 lut24; do
  echo "== c3 [$v]"; env MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_$v.so $C3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['us_per_launch'], d['roofline']['frac'])"
done > $O/c3_lut_inflight.txt 2>&1
cat $O/c3_lut_inflight.txt
MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_front.so timeout 250 python -u tests/tools/front_dev_counters.py 100000000 4096 2>&1 | grep "^k_" > $O/front.txt; cat $O/front.txt
scripts/ab_variants.sh "tests/tools/latency_trace.py 100000000 4096" knobs lut12 lut16 lut24 > $O/ab.txt 2>&1; cat $O/ab.txt
