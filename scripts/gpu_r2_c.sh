#!/bin/bash
# round 2, GPU call C: ablations + PMC of the flat GEMM variants; re-run of the failed tests
cd "$(dirname "$0")/.."
O=gpurun_out/r2c; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_encode.py tests/test_gpu_concurrency.py tests/test_gpu_train.py -m gpu -q --timeout=600 > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log
ROWS=4000000
for v in 2 4 6; do timeout 120 python tests/tools/flat_gemm_time.py $ROWS $v 2>&1 | tail -1; done | tee $O/gemm_times.txt
for v in 4 6; do timeout 120 python tests/tools/flat_gemm_time.py $ROWS $v 1 2>&1 | tail -1; done | tee -a $O/gemm_times.txt
for mask in 1 2 4 8 16 3 7; do
  for v in 4 6; do
    echo -n "abl$mask: "; MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_abl$mask.so timeout 200 python tests/tools/flat_gemm_time.py $ROWS $v 2>&1 | tail -1
  done
done | tee $O/ablations.txt
for v in 2 4 6; do scripts/pmc_flat.sh $ROWS $O/pmc_v$v $v > $O/pmc_v$v.log 2>&1; tail -4 $O/pmc_v$v.log; done
