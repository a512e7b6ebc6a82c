#!/bin/bash
# Round 6 dev A/B on an MI355X box: parity of the table-image path, then knob A/Bs at the reference's default shape
# usage: gpurun -- 'bash scripts/gpu_r06_ab.sh "spec spec ..." [nprobes]'   (spec = variant[:KNOB=v,KNOB=v])
O=gpurun_out/r06ab
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$O
cd $R
[ -n "$SKIP_PYTEST" ] || timeout 600 python -m pytest tests/test_gpu_lut_images.py -x -q > $O/pytest_lut.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_lut.txt
T="tests/tools/default_shape_time.py 100000000 768 48 ${2:-20} 2048"
export IMAGES_ONLY=1
for spec in $1; do
  v="${spec%%:*}"; kn=""; [ "$spec" != "$v" ] && kn="${spec#*:}"
  echo "== variant $v [$kn]"
  env ${kn//,/ } MI355_ANN_LIB=$R/lancedb_amd/variants/lib_$v.so timeout 300 python -u $T 2>&1 | grep "^nprobe"
done 2>&1 | tee $O/ab.txt
unset IMAGES_ONLY
if [ -n "$3" ]; then
  MI355_ANN_LIB=$R/lancedb_amd/variants/lib_dev.so timeout 300 python -u $T 2>&1 | grep "^nprobe" | tee $O/dev_counters.txt
  timeout 120 python -u tests/tools/write_bw_probe.py 2>&1 | grep "TB/s" | tee $O/write_bw.txt
fi
