#!/bin/bash
# Dev: run one tool per built variant / knob setting on the GPU box and print its result lines.
# usage: scripts/ab_variants.sh "tool and args" variant[:KNOB=val,KNOB=val] ...
#   e.g. scripts/ab_variants.sh "tests/tools/latency_trace.py" knobs knobs:MI355_PLAN_SPARSE=0 knobs:MI355_MERGE_BLOCK_MAX_NQ=0
# (knobs only act in variants built with -DMI355_DEV_KNOBS, scripts/build_variants.sh)
cd "$(dirname "$0")/.."
TOOL=$1; shift
for spec in "$@"; do
  v="${spec%%:*}"; kn=""; [ "$spec" != "$v" ] && kn="${spec#*:}"
  echo "== variant $v [${kn}]"
  env ${kn//,/ } MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_$v.so timeout 300 python -u $TOOL 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl"
done
