#!/bin/bash
# round 2, call N: the committed state as the driver runs it: whole GPU suite in one process, smoke(),
# the default bench line
O=gpurun_out/r2n
mkdir -p $O
S=$(date +%s)
timeout 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$? wall=$(( $(date +%s) - S ))s"; grep -v "amdgpu.ids" $O/pytest_gpu.log | tail -6
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?"; tail -1 $O/smoke.log
S=$(date +%s)
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$? wall=$(( $(date +%s) - S ))s"; tail -2 $O/bench_default.err; head -c 600 $O/bench_default.json; echo
