#!/bin/bash
# Round 6 dev run (gpurun -- 'bash scripts/gpu_r06_lut.sh'): parity of the batch-level distance tables, A/Bs of their knobs at the
# reference's default index shape, and the per-kernel times of one run (rocprofv3 --kernel-trace --stats).
O=gpurun_out/r06b
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$O
cd $R
timeout 600 python -m pytest tests/test_gpu_lut_images.py -x -q > $O/pytest_lut.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_lut.txt
T="tests/tools/default_shape_time.py 100000000 768 48 20 2048"
export IMAGES_ONLY=1
for spec in knobs knobs:MI355_LUT_WARM_AHEAD=0 knobs:MI355_LUT_WARM_AHEAD=4 knobs:MI355_LUT_WARM_AHEAD=16 nopf; do
  v="${spec%%:*}"; kn=""; [ "$spec" != "$v" ] && kn="${spec#*:}"
  echo "== variant $v [$kn]"
  env ${kn//,/ } MI355_ANN_LIB=$R/lancedb_amd/variants/lib_$v.so timeout 300 python -u $T 2>&1 | grep "^nprobe"
done 2>&1 | tee $O/ab.txt
unset IMAGES_ONLY
timeout 300 python -u $T 2>&1 | grep "^nprobe\|^rows" | tee $O/default_shape.txt
cd /tmp && export TMPDIR=/tmp
IMAGES_ONLY=1 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o s --output-format csv -- python $R/$T > $R/$O/prof.log 2>&1
echo "rocprof rc=$?"
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-220
find $O/prof -type f ! -name "*kernel_stats.csv" -delete
