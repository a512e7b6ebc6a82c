#!/bin/bash
# Dev: run the tune harness once per built variant (parity check via dbg_skew first)
for v in "$@"; do
  export MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_$v.so
  echo "== variant $v"
  M=96 NQ=64 timeout 120 python -u tests/tools/dbg_skew.py 2>&1 | grep "^M "
  timeout 200 python -u scripts/tune.py scripts/pmc_run.py 2>&1 | grep "default"
done
