#!/bin/bash
# Round-6 evidence run on an MI355X box (gpurun -- 'bash scripts/gpu_profiles_r06.sh [outdir]'):
#   rocprofv3 --kernel-trace --stats of the C3, default-shape (rows / 8192 partitions, m = dim / 16, nprobes 20) and flat C2 commands
#   (kernel averages the bench's own HIP events must agree with), and PMC passes of the scan kernels — each counter set in its OWN
#   run, never combined with sys / hip / hsa tracing.  The summaries kept are copied from the out directory into profiles/r06_* by hand.
O=${1:-gpurun_out/prof_r06}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
C3="python $R/bench.py --steps 5 --warmup 1 --recall-rows 0 --recall2-rows 0 --secondary 0 --cpu-seconds 0"
DS="python $R/tests/tools/default_shape_time.py 100000000 768 48 20 2048"
export IMAGES_ONLY=1
stats() { # name cmd...
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/$name -o s --output-format csv -- "$@" > $R/$O/$name.log 2>&1
  echo "stats $name rc=$?"
  find $R/$O/$name -type f ! -name "*kernel_stats.csv" -delete
}
pmc() { # name counters -- cmd...
  local name=$1; shift
  local ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
  shift
  timeout 400 rocprofv3 --kernel-trace --pmc "${ctr[@]}" -d $R/$O/$name -o p --output-format csv -- "$@" > $R/$O/$name.log 2>&1
  echo "pmc $name rc=$?"
  find $R/$O/$name -name "*kernel_trace.csv" -delete
}
LDS="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
stats c3 $C3
stats dflt $DS
stats flat python $R/bench.py --workload flat --steps 5 --warmup 1 --cpu-seconds 0
pmc c3_fetch FETCH_SIZE -- $C3 --steps 2
pmc c3_lds $LDS -- $C3 --steps 2
pmc dflt_fetch FETCH_SIZE -- $DS
pmc dflt_write WRITE_SIZE -- $DS
pmc dflt_lds $LDS -- $DS
# the latency paths: kernel traces of single queries / batches of 8 on the C3 index and on the reference's default shape (profiles/r06_o_*)
for cfg in "96 64 4096 1" "96 64 4096 8" "48 20 12207 1" "48 20 12207 8"; do
  set -- $cfg
  echo "=== m=$1 nprobe=$2 nlist=$3 batch=$4" >> $R/$O/latency_gaps.txt
  rm -rf /tmp/tr_lat
  LAT_M=$1 LAT_NPROBE=$2 timeout 400 rocprofv3 --kernel-trace -d /tmp/tr_lat -o t --output-format csv -- python -u $R/tests/tools/latency_trace.py 100000000 $3 $4 2>&1 | grep "per call" >> $R/$O/latency_gaps.txt
  python $R/scripts/trace_gaps.py /tmp/tr_lat k_coarse_lat >> $R/$O/latency_gaps.txt 2>&1
done
rm -rf /tmp/tr_lat
cd $R
for n in c3 dflt flat; do f=$(find $O/$n -name "*kernel_stats.csv" | head -1); echo "== $n $f"; head -9 "$f" | cut -c1-230; done
grep "^nprobe" $O/dflt.log | head -3
echo "--- k_scan_skew (C3: <96, ...>; default shape: <48, ..., true>)"; python scripts/pmc_summary.py $O k_scan_skew | tee $O/summary_scan.txt
echo "--- k_lut_images"; python scripts/pmc_summary.py $O k_lut_images | tee $O/summary_lut.txt
find $O -name "*counter_collection.csv" -size +4M -delete  # (the summaries above are what is kept)
