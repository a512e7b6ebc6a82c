#!/bin/bash
# round 2, call M: optimistic sixteen-wave selection for 128 < kk <= 256 (k_scan_skew OPT) against
# the shared-threshold build: same results required, scan time vs kk, the long-list tests
O=gpurun_out/r2m
mkdir -p $O
export MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_opt.so
timeout 300 python -m pytest tests/test_gpu_bigk.py -x -q -m gpu --timeout=250 -p no:cacheprovider > $O/pytest_bigk.log 2>&1
echo "pytest bigk rc=$?"; grep -v amdgpu.ids $O/pytest_bigk.log | tail -12
timeout 300 python tests/tools/scan_kk_time.py > $O/kk_opt.txt 2>&1
echo "opt rc=$?"; grep "^k " $O/kk_opt.txt
