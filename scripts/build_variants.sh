#!/bin/bash
# Dev: build kernel variants side by side (lancedb_amd/variants/lib_<name>.so) for A/B runs
# with MI355_ANN_LIB.  usage: scripts/build_variants.sh name:knob=val,knob=val ...
set -e
cd "$(dirname "$0")/.."
mkdir -p lancedb_amd/variants
for spec in "$@"; do
  name="${spec%%:*}"; knobs="${spec#*:}"
  python scripts/gen_skew_chunks.py ${knobs//,/ } > /dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
     -fPIC -shared -Wall -Wno-unused-function lancedb_amd/csrc/mi355_ann.hip -o lancedb_amd/variants/lib_$name.so
done

ls -la lancedb_amd/variants/
