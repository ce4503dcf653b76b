"""mpc_amd -- MI355X (gfx950) implementation of the per-party local arithmetic of victerying/mpc's
ABY3Engine and SecurennEngine.

The product is the C-ABI shared library ``mpc_amd/libmpcx.so`` (hand-written HIP kernels, declared in
``include/mpcx.h``) and the C++ host engines in ``mpc_amd/host`` that mirror the reference's classes on
top of it.  This Python package is only a thin ctypes binding used by the tests and by ``bench.py``;
there is no CPU fallback: importing :mod:`mpc_amd.lib` fails loudly when the library is missing."""
from .lib import Context, DeviceArray, MpcxError, load_library  # noqa: F401
