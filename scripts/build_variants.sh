#!/bin/bash
# Dev: build the library side by side with extra compile flags, for A/B runs selected with MI355_ANN_LIB
# (lancedb_amd/variants/ is git-ignored; gpurun ships it to the GPU box).
# usage: scripts/build_variants.sh name:FLAG[,FLAG...] ...
#   e.g. scripts/build_variants.sh knobs:-DMI355_DEV_KNOBS dev:-DMI355_DEV_KNOBS,-DMI355_DEV_COUNTERS ring:-DMI355_DEV_KNOBS,-DSK_RING_FULL=1
# -DMI355_DEV_KNOBS makes the library read the MI355_* environment knobs (the shipped library reads none);
# -DMI355_DEV_COUNTERS adds the per-phase counters tests/tools/scan_dev_counters.py prints.
set -e
cd "$(dirname "$0")/.."
mkdir -p lancedb_amd/variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  python - "$name" "$flags" <<'PY'
import sys
from lancedb_amd import _lib
name, flags = sys.argv[1], [f for f in sys.argv[2].split(",") if f]
_lib.build(extra_flags=flags, lib_path=f"lancedb_amd/variants/lib_{name}.so", obj_dir=f"/tmp/obj_{name}")
print("built", name, flags)
PY
done
ls -la lancedb_amd/variants/
