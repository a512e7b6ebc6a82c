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


class HostMappedArray:
    """A caller-owned host array, page-locked and mapped into the device's address space
    (hipHostRegister, mapped): `data_ptr()` is the DEVICE view, so the array can be handed to
    `IvfPqIndex.attach_raw_vectors` — the refine kernel then gathers its k * refine_factor rows
    per query over PCIe while the codes stay in HBM (C5: 100 M x 1536 raw vectors are 307 GB).
    The numpy array must outlive this object; `close()` (or the finaliser) unlocks the pages."""
    is_cuda = True

    def __init__(self, array, device=0):
        a = np.ascontiguousarray(array)
        if a is not array and a.base is not array:
            raise ValueError("HostMappedArray needs a C-contiguous array (it maps the caller's own pages)")
        self.host, self.shape, self.dtype, self.device, self.nbytes = a, a.shape, a.dtype, device, a.nbytes
        rt = runtime()
        rt.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
        rt.hipHostGetDevicePointer.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint]
        rt.hipHostUnregister.argtypes = [C.c_void_p]
        _check(rt.hipSetDevice(C.c_int(device)), "hipSetDevice")
        self._hp = C.c_void_p(a.ctypes.data)
        _check(rt.hipHostRegister(self._hp, C.c_size_t(a.nbytes), C.c_uint(2)), f"hipHostRegister({a.nbytes} B, mapped)")
        self._dp = C.c_void_p()
        err = rt.hipHostGetDevicePointer(C.byref(self._dp), self._hp, C.c_uint(0))
        if err != 0:
            rt.hipHostUnregister(self._hp)
            self._hp = None
            _check(err, "hipHostGetDevicePointer")

    def data_ptr(self):
        return self._dp.value or 0

    def close(self):
        if getattr(self, "_hp", None):
            runtime().hipHostUnregister(self._hp)
            self._hp = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def synchronize():
    _check(runtime().hipDeviceSynchronize(), "hipDeviceSynchronize")


class HostAllocArray:
    """A LIBRARY-allocated page-locked host array (hipHostMalloc, mapped, coherent): the other way a raw column can live in
    host memory — `HostMappedArray` registers pages the caller already owns (hipHostRegister).  Used by the C5 A/B of the
    PCIe gather (tests/tools/c5_gather_ab.py --c5-column hostmalloc): same rows, same kernel, the only difference is who
    allocated (and how the driver mapped) the pages.  `host` is a numpy view of the memory."""
    is_cuda = True

    def __init__(self, shape, dtype, device=0, flags=0):
        self.shape = tuple(int(x) for x in shape)
        self.dtype = np.dtype(dtype)
        self.device = device
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        rt = runtime()
        rt.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
        rt.hipHostFree.argtypes = [C.c_void_p]
        rt.hipHostGetDevicePointer.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint]
        _check(rt.hipSetDevice(C.c_int(device)), "hipSetDevice")
        self._hp = C.c_void_p()
        _check(rt.hipHostMalloc(C.byref(self._hp), C.c_size_t(max(self.nbytes, 16)), C.c_uint(flags)), f"hipHostMalloc({self.nbytes} B)")
        self._dp = C.c_void_p()
        _check(rt.hipHostGetDevicePointer(C.byref(self._dp), self._hp, C.c_uint(0)), "hipHostGetDevicePointer")
        buf = (C.c_char * max(self.nbytes, 16)).from_address(self._hp.value)
        self.host = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(self.shape, dtype=np.int64))).reshape(self.shape)

    def data_ptr(self):
        return self._dp.value or 0

    def close(self):
        if getattr(self, "_hp", None):
            self.host = None
            runtime().hipHostFree(self._hp)
            self._hp = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
