#!/bin/bash
# Dev: A/B of the flat GEMM schedules on one box.  usage: scripts/ab_flat.sh rows "name:lib:tile:persist" ...
# prints ms_per_step per variant and the group-minimum checksums (must all be equal)
ROWS=${1:-4000000}; shift
cd "$(dirname "$0")/.."
NEW=$PWD/lancedb_amd/libmi355_ann.so
for spec in "$@"; do
  IFS=: read name lib tile persist <<< "$spec"
  [ "$lib" = new ] && lib=$NEW || lib=$PWD/lancedb_amd/variants/lib_$lib.so
  out=$(MI355_ANN_LIB=$lib MI355_FLAT_TILE=$tile MI355_FLAT_PERSIST=$persist timeout 120 python bench.py --workload flat --flat-rows $ROWS --steps 6 --warmup 2 --cpu-seconds 0 2>&1 | tail -1)
  echo "$name $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), "ms", round(d["roofline"]["achieved"],1), "TF")' 2>/dev/null || echo "FAILED: $out")"
  MI355_ANN_LIB=$lib MI355_FLAT_TILE=$tile MI355_FLAT_PERSIST=$persist MI355_FLAT_SYNC=1 timeout 120 python bench.py --workload flat --flat-rows $ROWS --steps 1 --warmup 0 --cpu-seconds 0 2>&1 | grep checksum
done
