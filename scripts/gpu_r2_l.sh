#!/bin/bash
# round 2, call L: workgroup-shared q-th-best threshold in the production scan (kk > 64):
# scan time vs kk before / after (same results required), then the IVF-PQ parity tests on the new library
O=gpurun_out/r2l
mkdir -p $O
timeout 300 python tests/tools/scan_kk_time.py > $O/kk_before.txt 2>&1
echo "before rc=$?"; grep "^k " $O/kk_before.txt
MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_qshare.so timeout 300 python tests/tools/scan_kk_time.py > $O/kk_after.txt 2>&1
echo "after rc=$?"; grep "^k " $O/kk_after.txt
MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_qshare.so timeout 500 python -m pytest tests/test_gpu_bigk.py tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_gpu_concurrency.py -x -q -m gpu -k "not flat" --timeout=300 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -v amdgpu.ids $O/pytest.log | tail -8
