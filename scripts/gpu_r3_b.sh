#!/bin/bash
# round 3, call B: flat GEMM with the epilogue's operands staged by LDS-DMA (no global loads, no vmcnt(0) in the walk):
# parity of the flat path, then A/B against the previous library and the ablation builds, one box
O=gpurun_out/r3b
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "flat" > $O/pytest_flat.txt 2>&1
echo "pytest flat rc=$?"; tail -4 $O/pytest_flat.txt
R=4000000
for lib in base new; do
  [ $lib = new ] && L=$PWD/lancedb_amd/libmi355_ann.so || L=$PWD/lancedb_amd/variants/lib_$lib.so
  echo "== $lib"; MI355_ANN_LIB=$L timeout 300 python tests/tools/flat_gemm_time.py $R 4:1:l2 4:256:l2 4:512:l2 5:1:l2 4:1:cosine 4:256:cosine 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_$lib.txt
done
for abl in 1 4 8 16; do
  echo "== abl$abl"; MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_abl$abl.so timeout 200 python tests/tools/flat_gemm_time.py $R 4:1:l2 4:256:l2 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_abl.txt
done
echo "== full C2"; timeout 300 python tests/tools/flat_gemm_time.py 10000000 4:1:l2 4:256:l2 4:1:cosine 4:256:cosine 2>&1 | grep -v amdgpu.ids | tee $O/c2_new.txt
