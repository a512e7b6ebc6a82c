#!/bin/bash
# round 2, call I: native backtrace of the exit-time abort of the RCCL-using test process,
# before / after ncclCommFinalize in mi355_comm_destroy
O=gpurun_out/r2i
mkdir -p $O
export MI355_TEST_BACKTRACE=1
for i in 1 2; do
  MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_nofinalize.so timeout 150 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu -p no:cacheprovider > $O/old_$i.log 2>&1
  echo "old lib run $i rc=$?"; grep -v "amdgpu.ids" $O/old_$i.log | tail -45
done
for i in 1 2 3; do
  timeout 150 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu -p no:cacheprovider > $O/new_$i.log 2>&1
  echo "new lib run $i rc=$?"; grep -v "amdgpu.ids" $O/new_$i.log | tail -45
done
