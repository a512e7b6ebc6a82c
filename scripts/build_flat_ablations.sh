#!/bin/bash
# Dev: side-by-side libraries of the flat GEMM with parts ablated (kernels_flat_mfma8.h
# MI355_FLAT_ABLATE masks).  usage: scripts/build_flat_ablations.sh 1 2 4 8 16 ...
# -> lancedb_amd/variants/lib_abl<mask>.so (select with MI355_ANN_LIB; results are WRONG by design)
set -e
cd "$(dirname "$0")/.."
python -c "from lancedb_amd import _lib; _lib.build()"
mkdir -p lancedb_amd/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -Wall -Wno-unused-function"
OBJS=$(ls lancedb_amd/build/*.o | grep -v ann_flat.o)
for mask in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -DMI355_FLAT_ABLATE=$mask -c lancedb_amd/csrc/ann_flat.hip -o lancedb_amd/variants/ann_flat_abl$mask.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC $OBJS lancedb_amd/variants/ann_flat_abl$mask.o -shared -ldl -Wl,-rpath,/opt/rocm/lib -o lancedb_amd/variants/lib_abl$mask.so &&
    echo "built abl$mask" ) &
done
wait
