#!/bin/bash
# round 3, call J: latency mode (sliced work items, pinned staging) + the whole suite, then the default bench
O=gpurun_out/r3j
mkdir -p $O
export PYTHONFAULTHANDLER=1
S=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$? wall=$(( $(date +%s) - S ))s"; tail -6 $O/pytest.txt
S=$(date +%s)
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$? wall=$(( $(date +%s) - S ))s"; tail -5 $O/bench_default.err | cut -c1-400
