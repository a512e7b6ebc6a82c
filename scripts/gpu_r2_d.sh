#!/bin/bash
# round 2, GPU call D: the 4-slot flat GEMM: flat tests, timing at 4 M and 10 M rows
cd "$(dirname "$0")/.."
O=gpurun_out/r2d; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "flat" --timeout=600 > $O/pytest_flat.log 2>&1
echo "pytest flat rc=$?"; tail -6 $O/pytest_flat.log
for v in 2 4 7 8; do timeout 120 python tests/tools/flat_gemm_time.py 4000000 $v 2>&1 | tail -1; done | tee $O/gemm_times_4m.txt
for v in 4 7; do timeout 120 python tests/tools/flat_gemm_time.py 4000000 $v 1 2>&1 | tail -1; done | tee -a $O/gemm_times_4m.txt
for v in 2 7; do for m in l2 cosine; do timeout 200 python tests/tools/flat_gemm_time.py 10000000 $v 0 $m 2>&1 | tail -1; done; done | tee $O/gemm_times_10m.txt
scripts/pmc_flat.sh 4000000 $O/pmc_v7 7 k_flat_gemm > $O/pmc_v7.log 2>&1; tail -4 $O/pmc_v7.log
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_lance_loader.py tests/test_gpu_pq4.py -m gpu -q --timeout=600 > $O/pytest_misc.log 2>&1
echo "pytest misc rc=$?"; tail -6 $O/pytest_misc.log
