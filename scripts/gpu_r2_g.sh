#!/bin/bash
# round 2, call G: backtrace of the exit-time abort of the sharded test process; probe cases; large-m test
O=gpurun_out/r2g
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 400 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint" -ex run -ex "bt 40" \
  --args python -m pytest tests/test_gpu_sharded.py -x -q -m gpu -p no:cacheprovider > $O/gdb_sharded.log 2>&1
echo "gdb rc=$?"; grep -v "^\[New Thread\|^\[Thread\|^warning:\|LWP" $O/gdb_sharded.log | tail -60
timeout 400 python tests/tools/rccl_exit_probe.py > $O/exit_probe.txt 2>&1
echo "exit probe rc=$?"; cat $O/exit_probe.txt
timeout 300 python -m pytest tests/test_gpu_bigm.py -x -q -m gpu --timeout=200 -p no:cacheprovider > $O/pytest_bigm.log 2>&1
echo "pytest bigm rc=$?"; tail -15 $O/pytest_bigm.log
