#!/bin/bash
# round 3, call D: selection rework (nearest partition first, wave-parallel block merge, per-item counts,
# overlapped tail of the last pass): whole GPU suite, then scan time vs kk against the previous library
O=gpurun_out/r3d
mkdir -p $O
export PYTHONFAULTHANDLER=1
S=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$? wall=$(( $(date +%s) - S ))s"; tail -15 $O/pytest.txt
for lib in base new; do
  [ $lib = new ] && L=$PWD/lancedb_amd/libmi355_ann.so || L=$PWD/lancedb_amd/variants/lib_$lib.so
  echo "== $lib"; MI355_ANN_LIB=$L timeout 300 python tests/tools/scan_kk_time.py 25000000 1024 2>&1 | grep -v amdgpu.ids | tee $O/kk_$lib.txt
done
echo "== dev counters"; MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_dev.so timeout 300 python tests/tools/scan_dev_counters.py 25000000 1024 2>&1 | grep -v amdgpu.ids | tee $O/scan_dev_counters.txt
