#!/bin/bash
# round 2, call Q: whole GPU suite on the committed library (after the kk > 128 selection change)
O=gpurun_out/r2q
mkdir -p $O
S=$(date +%s)
timeout 600 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$? wall=$(( $(date +%s) - S ))s"; grep -v "amdgpu.ids" $O/pytest_gpu.log | tail -4
