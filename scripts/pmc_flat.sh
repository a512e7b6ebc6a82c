#!/bin/bash
# Dev: PMC passes over the flat GEMM (one rocprofv3 run per counter set; never combined with
# sys/hip/hsa tracing).  usage: scripts/pmc_flat.sh rows outdir gemm_variant [kernel substring] [metric]
ROWS=${1:-4000000}; OUT=${2:-gpurun_out/pmc_flat_r}; VAR=${3:-0}; KERN=${4:-k_flat_gemm}; METRIC=${5:-l2}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p $OUT
pass() { # name counters...
  local name=$1; shift
  timeout 150 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o p --output-format csv -- \
    python bench.py --workload flat --flat-rows $ROWS --steps 2 --warmup 1 --cpu-seconds 0 --flat-gemm $VAR --flat-metric $METRIC > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_MISC
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
python scripts/pmc_summary.py $OUT $KERN | tee $OUT/summary.txt
