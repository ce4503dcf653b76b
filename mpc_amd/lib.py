"""ctypes binding of libmpcx.so (the C ABI in include/mpcx.h).  Device memory is either owned by a
:class:`DeviceArray` (mpcx_malloc) or borrowed from a torch tensor via its ``data_ptr()``."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPCX_LIBRARY") or os.path.join(_HERE, "libmpcx.so")  # override: A/B runs of two builds

_lib = None


class MpcxError(RuntimeError):
    pass


def load_library():
    """Loads libmpcx.so.  There is no fallback: a missing library is an error."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MpcxError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        # A process that also uses PyTorch must bring torch's bundled HIP runtime up first: once this library (linked
        # against the system ROCm) has initialised HIP, torch no longer finds the device.
        import sys
        torch = sys.modules.get("torch")
        if torch is not None and torch.cuda.is_available():
            torch.cuda.init()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.mpcx_last_error.restype = ctypes.c_char_p
        _lib.mpcx_last_error.argtypes = [ctypes.c_void_p]
        _lib.mpcx_get_stream.restype = ctypes.c_void_p
    return _lib


def exported_symbols():
    """Names declared in include/mpcx.h (parsed from the header text)."""
    import re
    hdr = open(os.path.join(os.path.dirname(_HERE), "include", "mpcx.h")).read()
    return sorted(set(re.findall(r"\b(mpcx_[a-z0-9_]+)\s*\(", hdr)))


class DeviceArray:
    """A typed device buffer.  ``ptr`` is the raw device address."""

    def __init__(self, ctx, dtype, n, ptr=None, owner=None):
        self.ctx, self.dtype, self.n = ctx, np.dtype(dtype), int(n)
        self.nbytes = self.n * self.dtype.itemsize
        self._own = ptr is None
        self._owner = owner  # keeps a borrowed allocation (e.g. a torch tensor) alive
        if ptr is None:
            p = ctypes.c_void_p()
            ctx._call("mpcx_malloc", ctypes.c_uint64(max(self.nbytes, 8)), ctypes.byref(p))
            ptr = p.value
        self.ptr = ptr

    @property
    def size(self):
        return self.n

    def view(self, offset_elems, n, dtype=None):
        dt = np.dtype(dtype) if dtype is not None else self.dtype
        return DeviceArray(self.ctx, dt, n, ptr=self.ptr + offset_elems * self.dtype.itemsize, owner=self)

    def to_host(self):
        out = np.empty(self.n, dtype=self.dtype)
        if self.n:
            self.ctx._call("mpcx_d2h", ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(self.ptr),
                           ctypes.c_uint64(self.nbytes))
        return out

    def free(self):
        if self._own and self.ptr:
            if self.ctx.h:  # a closed context has already released every block it handed out
                self.ctx._call("mpcx_free", ctypes.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _a(x):
    if x is None:
        return ctypes.c_void_p(0)
    if isinstance(x, DeviceArray):
        return ctypes.c_void_p(x.ptr)
    if isinstance(x, (int, np.integer)):
        return ctypes.c_uint64(int(x) & 0xFFFFFFFFFFFFFFFF)
    if hasattr(x, "data_ptr"):  # torch tensor on the device
        return ctypes.c_void_p(x.data_ptr())
    raise TypeError(f"cannot pass {type(x)} to the C ABI")


class Context:
    """One mpcx context (one HIP stream, one scratch workspace) on one GPU."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = ctypes.c_void_p()
        rc = self.lib.mpcx_init(ctypes.c_int(device), ctypes.byref(h))
        if rc:
            raise MpcxError(self.lib.mpcx_last_error(None).decode())
        self.h = h
        self.device = device

    def close(self):
        if self.h:
            self.lib.mpcx_shutdown(self.h)
            self.h = None

    def _call(self, name, *args):
        if not self.h:
            raise MpcxError(f"{name}: the context is closed")
        rc = getattr(self.lib, name)(self.h, *args)
        if rc:
            raise MpcxError(f"{name}: {self.lib.mpcx_last_error(self.h).decode()}")

    def call(self, name, *args):
        """Calls ``mpcx_<name>(ctx, *args)``; DeviceArrays / tensors / ints are converted."""
        self._call("mpcx_" + name, *[_a(x) for x in args])

    def sync(self):
        self._call("mpcx_sync")

    def stream(self):
        return self.lib.mpcx_get_stream(self.h)

    def empty(self, dtype, n):
        return DeviceArray(self, dtype, n)

    def to_device(self, arr, dtype=None):
        arr = np.ascontiguousarray(arr, dtype=dtype)
        d = DeviceArray(self, arr.dtype, arr.size)
        if arr.size:
            self._call("mpcx_h2d", ctypes.c_void_p(d.ptr), ctypes.c_void_p(arr.ctypes.data), ctypes.c_uint64(arr.nbytes))
        return d

    def wrap(self, tensor, dtype):
        """Borrows a torch CUDA tensor's storage as a DeviceArray of `dtype`."""
        nbytes = tensor.numel() * tensor.element_size()
        return DeviceArray(self, dtype, nbytes // np.dtype(dtype).itemsize, ptr=tensor.data_ptr(), owner=tensor)
