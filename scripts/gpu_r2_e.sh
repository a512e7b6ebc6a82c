#!/bin/bash
# round 2, call E: which of the sharded / 4-bit / loader tests hangs (per-test timeouts, stack dump),
# flat GEMM schedules on clean inputs (census + fallback count), new flat tests
O=gpurun_out/r2e
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 240 python -m pytest tests/test_gpu_sharded.py -x -v -m gpu --timeout=70 --timeout-method=thread -p no:cacheprovider > $O/pytest_sharded.log 2>&1
echo "pytest sharded rc=$?"
tail -60 $O/pytest_sharded.log
timeout 300 python -m pytest tests/test_gpu_pq4.py tests/test_lance_loader.py -x -v -m gpu --timeout=100 --timeout-method=thread -p no:cacheprovider > $O/pytest_pq4_loader.log 2>&1
echo "pytest pq4+loader rc=$?"
tail -40 $O/pytest_pq4_loader.log
timeout 200 python tests/tools/flat_gemm_time.py 4000000 2:0:l2 4:0:l2 4:1:l2 7:0:l2 7:1:l2 8:0:l2 1:0:l2 2:0:cosine 4:1:cosine 7:1:cosine > $O/gemm_4m.txt 2>&1
echo "gemm 4m rc=$?"; cat $O/gemm_4m.txt | tail -14
timeout 240 python tests/tools/flat_gemm_time.py 10000000 2:0:l2 4:0:l2 4:1:l2 7:1:l2 2:0:cosine 4:1:cosine 7:1:cosine > $O/gemm_10m.txt 2>&1
echo "gemm 10m rc=$?"; cat $O/gemm_10m.txt | tail -10
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "flat" --timeout=250 -p no:cacheprovider > $O/pytest_flat.log 2>&1
echo "pytest flat rc=$?"; tail -15 $O/pytest_flat.log
