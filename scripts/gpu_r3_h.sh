#!/bin/bash
# round 3, call H: the whole GPU suite on the final selection code, then the default bench line end to end
# (new legs: secondary.loopback_world8, secondary.c5_refine10) with its wall time
O=gpurun_out/r3h
mkdir -p $O
export PYTHONFAULTHANDLER=1
S=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
echo "pytest rc=$? wall=$(( $(date +%s) - S ))s"; tail -6 $O/pytest.txt
S=$(date +%s)
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$? wall=$(( $(date +%s) - S ))s"; tail -5 $O/bench_default.err; head -c 20000 $O/bench_default.json; echo
