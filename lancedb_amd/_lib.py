"""Loader / builder of the native engine (libmi355_ann.so).

There is no CPU fallback: if the HIP library is missing or no gfx950 device is
visible, every search call raises.  The library is built in-tree with hipcc
(cross-compiles without a GPU) so it travels with the repository snapshot.
"""
import ctypes as C
import os
import subprocess

from . import _abi

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)
# MI355_ANN_LIB: dev override used to A/B kernel variants built side by side
LIB_PATH = os.environ.get("MI355_ANN_LIB") or os.path.join(_PKG, "libmi355_ann.so")
_CSRC = os.path.join(_PKG, "csrc")
# the translation unit first, then everything it includes (any change rebuilds)
_SOURCES = [os.path.join(_CSRC, "mi355_ann.hip")] + sorted(
    os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".h", ".inc")))
_HEADER = os.path.join(ROOT, "include", "mi355_ann.h")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
               "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC", "-shared", "-Wall",
               "-Wno-unused-function"]

_lib = None


class EngineError(RuntimeError):
    """Mirrors lancedb::Error variants (rust/lancedb/src/error.rs:55-145)."""

    def __init__(self, status, message):
        self.status = status
        name = {1: "InvalidInput", 2: "Runtime", 3: "Timeout", 4: "NotSupported"}.get(status, str(status))
        super().__init__(f"{name}: {message}")


class InvalidInput(EngineError, ValueError):
    pass


class NotSupported(EngineError):
    pass


class QueryTimeout(EngineError, TimeoutError):
    pass


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in _SOURCES + [_HEADER])


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 the C-ABI library in-tree."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc] + HIPCC_FLAGS + [_SOURCES[0], "-o", LIB_PATH]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed building libmi355_ann.so:\n" + r.stdout[-4000:])
    return LIB_PATH


def _bind_hip_runtime():
    """One HIP runtime per process.  libmi355_ann.so needs `libamdhip64.so.7`.
    A PyTorch-ROCm wheel ships its own copy with that SONAME but loads it under
    the name `libamdhip64.so`, so if the engine pulled in /opt/rocm's copy first
    a later `import torch` would start a SECOND runtime and fail with "No HIP
    GPUs are available".  When torch is installed (bench.py / torch.distributed
    plumbing), pre-load its copy so the engine's NEEDED entry resolves to it by
    SONAME; otherwise the system runtime is used.  MI355_HIP_RUNTIME=system opts out."""
    if os.environ.get("MI355_HIP_RUNTIME", "auto") == "system":
        return
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # torch already loaded its runtime; the SONAME match reuses it
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """The loaded C-ABI library.  Raises if it has not been built (never falls
    back to anything else)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  The MI355X engine has no CPU fallback.")
        _bind_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.mi355_abi_version.restype = C.c_uint32
        for name in _abi.EXPORTED_SYMBOLS:
            if name != "mi355_abi_version":
                getattr(L, name).restype = C.c_int32
        if L.mi355_abi_version() != _abi.ABI_VERSION:
            raise ImportError("libmi355_ann.so ABI version mismatch; rebuild")
        _lib = L
    return _lib


def last_error():
    buf = C.create_string_buffer(1024)
    lib().mi355_last_error(buf, C.c_size_t(1024))
    return buf.value.decode("utf-8", "replace")


def check(status):
    if status == _abi.OK:
        return
    msg = last_error()
    cls = {_abi.ERR_INVALID_INPUT: InvalidInput, _abi.ERR_NOT_SUPPORTED: NotSupported,
           _abi.ERR_TIMEOUT: QueryTimeout}.get(status, EngineError)
    raise cls(status, msg)


def device_count():
    n = C.c_int32(0)
    check(lib().mi355_device_count(C.byref(n)))
    return n.value
