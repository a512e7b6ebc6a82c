#!/bin/bash
# ASan + UBSan build of the HOST side of libmi355_ann.so (device code is compiled as usual) and the CPU-only ABI
# tests under it: descriptor / parameter validation, struct sizes, exports, error slot — everything that runs
# before a device is touched (sanitizer row).  Needs no GPU.
set -e
cd "$(dirname "$0")/.."
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
from lancedb_amd import _lib
san = ["-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-omit-frame-pointer", "-Xarch_host", "-fno-sanitize-recover=undefined", "-g"]
_lib._link_flags_orig = _lib._link_flags
_lib._link_flags = lambda: _lib._link_flags_orig() + ["-fsanitize=address,undefined", "-shared-libsan"]
print(_lib.build(extra_flags=san, lib_path=os.path.join(_lib._PKG, "variants", "lib_asan.so"),
                 obj_dir=os.path.join(_lib._PKG, "variants", "obj_asan")))
PY
export MI355_ANN_LIB=$PWD/lancedb_amd/variants/lib_asan.so
export LD_PRELOAD=$RT
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:verify_asan_link_order=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python -m pytest tests/test_abi.py tests/test_build_host.py -x -q -m "not gpu" "$@"
