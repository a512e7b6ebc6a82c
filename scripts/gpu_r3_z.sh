#!/bin/bash
# round 3, call Z: does a deeper code ring shorten the scan phase of sliced (single-query) work items?
for v in dev ring; do
  echo "== lib_$v"
  MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_$v.so timeout 200 python tests/tools/scan_dev_counters.py 25000000 1 2>&1 | grep "^k 10"
  MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_$v.so timeout 200 python tests/tools/scan_dev_counters.py 25000000 1024 2>&1 | grep "^k 10"
  MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_$v.so timeout 120 python tests/tools/latency_trace.py 2>&1 | grep "single query"
done
