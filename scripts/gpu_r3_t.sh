#!/bin/bash
# round 3, call T: PMC passes of the flat GEMM at the full C2 shape, both metrics (own rocprofv3 passes, kernel trace only)
O=gpurun_out/r3t
mkdir -p $O
bash scripts/pmc_flat.sh 10000000 $O/pmc_l2 0 k_flat_gemm8 l2 2>&1 | tail -5
bash scripts/pmc_flat.sh 10000000 $O/pmc_cosine 0 k_flat_gemm8 cosine 2>&1 | tail -5
find $O -name "*kernel_trace.csv" -size +5M -delete; find $O -name "*.csv" -size +30M -delete
