#!/bin/bash
# round 2, call F: exit behaviour of RCCL-using processes; the default bench line end to end;
# rocprofv3 kernel stats of the C3 and flat C2 benches
O=gpurun_out/r2f
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 400 python tests/tools/rccl_exit_probe.py > $O/exit_probe.txt 2>&1
echo "exit probe rc=$?"; cat $O/exit_probe.txt
S=$(date +%s)
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$? wall=$(( $(date +%s) - S ))s"; tail -3 $O/bench_default.err; head -c 6000 $O/bench_default.json; echo
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c3 -o c3 -- python $R/bench.py --steps 5 --warmup 1 --recall-rows 0 --secondary 0 --cpu-seconds 0 > $R/$O/prof_c3.log 2>&1
echo "rocprof c3 rc=$?"; tail -2 $R/$O/prof_c3.log | head -c 1500; echo
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_flat -o flat -- python $R/bench.py --workload flat --steps 5 --warmup 1 --cpu-seconds 0 > $R/$O/prof_flat.log 2>&1
echo "rocprof flat rc=$?"; tail -2 $R/$O/prof_flat.log | head -c 1500; echo
cd $R
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*_kernel_stats.csv" | head
