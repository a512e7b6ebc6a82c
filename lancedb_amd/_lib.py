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
_OBJ = os.path.join(_PKG, "build")
# one object per translation unit (compiled in parallel); every header / generated include is
# a dependency of every unit
_UNITS = sorted(f for f in os.listdir(_CSRC) if f.endswith(".hip"))
_HEADERS = sorted(os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".h", ".inc")))
_HEADER = os.path.join(ROOT, "include", "mi355_ann.h")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
               "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    hipcc = os.environ.get("HIPCC") or os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
    return hipcc if os.path.exists(hipcc) else "hipcc"


def _link_flags():
    """The library links the HIP runtime only (hipcc adds it); RCCL — the multi-GPU exchange behind
    mi355_comm_* — is dlopen'ed by the first communicator call.  The rpath is the lib directory of the
    ROCm installation the compiler belongs to, not a hard-coded one."""
    import shutil
    exe = shutil.which(_hipcc()) or _hipcc()
    rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(exe))), "lib")
    flags = ["-shared", "-ldl"]
    if os.path.isdir(rocm_lib):
        flags.append("-Wl,-rpath," + rocm_lib)
    return flags


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


def _newer(path, deps):
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _stale():
    srcs = [os.path.join(_CSRC, u) for u in _UNITS]
    return _newer(LIB_PATH, srcs + _HEADERS + [_HEADER])


def build(force=False, verbose=False, extra_flags=(), lib_path=None, obj_dir=None):
    """hipcc --offload-arch=gfx950 the C-ABI library in-tree: one object per unit of
    csrc/*.hip (in parallel), then one link.  `extra_flags` / `lib_path` / `obj_dir`
    build a side-by-side dev variant (scripts/build_variants.sh)."""
    lib_path = lib_path or LIB_PATH
    obj_dir = obj_dir or _OBJ
    if not force and not extra_flags and lib_path == LIB_PATH and not _stale():
        return lib_path
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for u in _UNITS:
        src = os.path.join(_CSRC, u)
        obj = os.path.join(obj_dir, u[:-4] + ".o")
        if force or extra_flags or _newer(obj, [src] + _HEADERS + [_HEADER]):
            cmd = [hipcc] + HIPCC_FLAGS + list(extra_flags) + ["-c", src, "-o", obj]
            jobs.append((u, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = []
    for u, p in jobs:
        out, _ = p.communicate()
        if verbose or p.returncode != 0:
            print(out)
        if p.returncode != 0:
            failed.append((u, out))
    if failed:
        raise RuntimeError("hipcc failed building libmi355_ann.so:\n" +
                           "\n".join(f"--- {u}\n{out[-4000:]}" for u, out in failed))
    objs = [os.path.join(obj_dir, u[:-4] + ".o") for u in _UNITS]
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC"] + objs + _link_flags() + ["-o", lib_path]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0:
        raise RuntimeError("link of libmi355_ann.so failed:\n" + r.stdout[-4000:])
    return lib_path


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
    # the same holds for RCCL (SONAME librccl.so.1 in both copies): one collective library per process
    for name in ("libamdhip64.so", "librccl.so"):
        cand = os.path.join(os.path.dirname(spec.origin), "lib", name)
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
