#!/bin/bash
# round 2, call J: exit probe (torch / RCCL order), the whole GPU suite in one process as the driver
# runs it, smoke(), the default bench line
O=gpurun_out/r2j
mkdir -p $O
timeout 300 python tests/tools/rccl_exit_probe.py > $O/exit_probe.txt 2>&1
echo "exit probe rc=$?"; cat $O/exit_probe.txt
S=$(date +%s)
timeout 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$? wall=$(( $(date +%s) - S ))s"; grep -v "amdgpu.ids" $O/pytest_gpu.log | tail -12
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?"; tail -2 $O/smoke.log
S=$(date +%s)
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$? wall=$(( $(date +%s) - S ))s"; tail -3 $O/bench_default.err; head -c 1200 $O/bench_default.json; echo
