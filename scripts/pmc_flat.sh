#!/bin/bash
# Dev: PMC passes over the flat GEMM (one rocprofv3 run per counter set; never combined with
# sys/hip/hsa tracing).  usage: scripts/pmc_flat.sh rows outdir  ->  outdir/<set>/p_counter_collection.csv
ROWS=${1:-4000000}; OUT=${2:-gpurun_out/pmc_flat_r}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 -L > $OUT/counters.txt 2>&1
pass() { # name counters...
  local name=$1; shift
  timeout 150 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o p --output-format csv -- \
    python bench.py --workload flat --flat-rows $ROWS --steps 2 --warmup 1 --cpu-seconds 0 > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM
pass sq3 SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pass ta TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
python scripts/pmc_summary.py $OUT k_flat_gemm | tee $OUT/summary.txt
