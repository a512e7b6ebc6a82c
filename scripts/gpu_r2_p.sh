#!/bin/bash
# round 2, call P: optimistic sixteen-wave passes for every kk > 128: long-list tests, scan time vs kk
O=gpurun_out/r2p
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_bigk.py -x -q -m gpu --timeout=250 -p no:cacheprovider > $O/pytest_bigk.log 2>&1
echo "pytest bigk rc=$?"; grep -v amdgpu.ids $O/pytest_bigk.log | tail -12
timeout 300 python tests/tools/scan_kk_time.py > $O/kk.txt 2>&1
echo "kk rc=$?"; grep "^k " $O/kk.txt
