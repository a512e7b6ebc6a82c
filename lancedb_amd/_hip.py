"""Minimal HIP-runtime binding (ctypes) for device buffers.

The engine's C ABI takes raw device pointers, so callers can bring any device
memory (torch tensors, hip-python, cupy ...).  `DeviceArray` is the smallest
self-contained option: tests and tools use it so that nothing needs PyTorch.
"""
import ctypes as C

import numpy as np

from ._lib import EngineError, lib

_rt = None
_H2D, _D2H, _D2D = 1, 2, 3


def runtime():
    global _rt
    if _rt is None:
        lib()  # loads libmi355_ann.so, which pulls in (or re-uses) the HIP runtime
        for name in ("libamdhip64.so.7", "libamdhip64.so"):
            try:
                _rt = C.CDLL(name)
                break
            except OSError:
                continue
        if _rt is None:
            raise ImportError("HIP runtime (libamdhip64) not found")
        _rt.hipGetErrorString.restype = C.c_char_p
    return _rt


def _check(err, what):
    if err != 0:
        raise EngineError(2, f"{what}: {runtime().hipGetErrorString(err).decode()}")


class DeviceArray:
    """A typed device allocation with numpy round-trips; exposes `data_ptr()` /
    `is_cuda` like a torch tensor so the index classes accept either."""
    is_cuda = True

    def __init__(self, shape, dtype, device=0):
        self.shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.device = device
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self._p = C.c_void_p()
        rt = runtime()
        _check(rt.hipSetDevice(C.c_int(device)), "hipSetDevice")
        _check(rt.hipMalloc(C.byref(self._p), C.c_size_t(max(self.nbytes, 16))), "hipMalloc")

    @classmethod
    def from_numpy(cls, a, device=0):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype, device)
        if a.nbytes:
            _check(runtime().hipMemcpy(d._p, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), C.c_int(_H2D)),
                   "hipMemcpy H2D")
        return d

    def numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        rt = runtime()
        _check(rt.hipDeviceSynchronize(), "hipDeviceSynchronize")
        if self.nbytes:
            _check(rt.hipMemcpy(out.ctypes.data_as(C.c_void_p), self._p, C.c_size_t(self.nbytes), C.c_int(_D2H)),
                   "hipMemcpy D2H")
        return out

    def copy_from(self, src):
        """Device-to-device copy of another device array of the same byte size."""
        if self.nbytes:
            _check(runtime().hipMemcpy(self._p, C.c_void_p(src.data_ptr()), C.c_size_t(self.nbytes), C.c_int(_D2D)),
                   "hipMemcpy D2D")
        return self

    def data_ptr(self):
        return self._p.value or 0

    def contiguous(self):
        return self

    def view(self, *shape):
        n = int(np.prod(self.shape, dtype=np.int64))
        shape = list(shape)
        if -1 in shape:
            known = int(np.prod([s for s in shape if s != -1], dtype=np.int64))
            shape[shape.index(-1)] = n // max(known, 1)
        v = object.__new__(DeviceArray)
        v.shape, v.dtype, v.device, v.nbytes, v._p, v._base = tuple(shape), self.dtype, self.device, self.nbytes, self._p, self
        return v

    def free(self):
        if getattr(self, "_base", None) is None and self._p:
            runtime().hipFree(self._p)
        self._p = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def synchronize():
    _check(runtime().hipDeviceSynchronize(), "hipDeviceSynchronize")
