#!/bin/bash
# Round-4 evidence run on an MI355X box (gpurun -- 'bash scripts/gpu_profiles_r04.sh [outdir]'):
#   rocprofv3 --kernel-trace --stats of the C3, flat C2 and C4 commands (kernel averages the bench's own HIP events
#   must agree with), and PMC passes — each counter set in its OWN run, never combined with sys / hip / hsa tracing —
#   of the scan kernel at kk = 10, kk = 250 and batch 256 and of the shipped flat GEMM (SCAN_ONLY=1 skips the flat passes).
# The summaries a round keeps are copied from the out directory into profiles/ by hand (see profiles/r04_*).
O=${1:-gpurun_out/prof_r04}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
C3="python $R/bench.py --steps 5 --warmup 1 --recall-rows 0 --recall2-rows 0 --secondary 0 --cpu-seconds 0"
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
[ -z "$SCAN_ONLY" ] && stats flat python $R/bench.py --workload flat --steps 5 --warmup 1 --cpu-seconds 0
stats c4 python $R/bench.py --workload c4 --loopback-world 0 --cpu-seconds 0 --steps 6
pmc scan_kk10_fetch FETCH_SIZE -- $C3 --steps 2
pmc scan_kk10_lds $LDS -- $C3 --steps 2
pmc scan_kk250_fetch FETCH_SIZE -- $C3 --steps 2 --k 250
pmc scan_kk250_lds $LDS -- $C3 --steps 2 --k 250
pmc scan_b256_fetch FETCH_SIZE -- $C3 --steps 4 --batch 256
FL="python $R/bench.py --workload flat --steps 2 --warmup 1 --cpu-seconds 0"
if [ -z "$SCAN_ONLY" ]; then
pmc flat_fetch FETCH_SIZE -- $FL
pmc flat_sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- $FL
pmc flat_grbm GRBM_GUI_ACTIVE GRBM_COUNT -- $FL
fi
cd $R
for n in c3 c4 flat; do f=$(find $O/$n -name "*kernel_stats.csv" | head -1); echo "== $n $f"; head -8 "$f" | cut -c1-220; done
du -sh $O/* | sort -h | tail -5
python scripts/pmc_summary.py $O k_scan_skew | tee $O/summary_scan.txt
[ -z "$SCAN_ONLY" ] && python scripts/pmc_summary.py $O k_flat_gemm8 | tee $O/summary_flat.txt
find $O -name "*counter_collection.csv" -size +4M -delete  # (the summaries above are what is kept)
