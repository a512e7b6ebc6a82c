#!/bin/bash
# round 3, call G: from which kk on should every query's nearest partition run first? (knob build, one box)
O=gpurun_out/r3g
mkdir -p $O
for bf in 65 33 1; do
  echo "== best-first from kk >= $bf"; MI355_BEST_FIRST_MIN_KK=$bf MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_knobs.so timeout 300 python tests/tools/scan_kk_time.py 25000000 1024 2>&1 | grep -v amdgpu.ids | tee $O/kk_bf$bf.txt
done
echo "== base"; MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_base.so timeout 300 python tests/tools/scan_kk_time.py 25000000 1024 2>&1 | grep -v amdgpu.ids | tee $O/kk_base.txt
