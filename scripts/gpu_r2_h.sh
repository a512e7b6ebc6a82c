#!/bin/bash
# round 2, call H: narrow the exit-time heap corruption of local-array shard handles
O=gpurun_out/r2h
mkdir -p $O
timeout 500 python tests/tools/rccl_exit_probe.py > $O/exit_probe.txt 2>&1
echo "exit probe rc=$?"; cat $O/exit_probe.txt
