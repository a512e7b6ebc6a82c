#!/bin/bash
# round 2, call R: the default bench line on the final committed library
O=gpurun_out/r2r
mkdir -p $O
timeout 230 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?"; head -c 400 $O/bench_default.json; echo
