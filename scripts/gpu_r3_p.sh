#!/bin/bash
# round 3, call P: whole GPU suite + the default bench line end to end (wall time printed) + rocprofv3 kernel stats
O=gpurun_out/r3p
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
