"""CPU-side checks of the C ABI: the library loads and exports every symbol include/mpcx.h declares,
and the product binding has no fallback path (no compute is attempted without a GPU)."""
import ctypes
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from mpc_amd.lib import exported_symbols, load_library
    lib = load_library()
    names = exported_symbols()
    assert len(names) > 60
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/mpcx.h but not exported: {missing}"


def test_init_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from mpc_amd import Context, MpcxError
    with pytest.raises(MpcxError):
        Context(0)


def test_product_never_references_the_oracle():
    """The oracle is test infrastructure: nothing under mpc_amd/ or include/ may mention it."""
    bad = []
    for base in ("mpc_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if "liboracle" in txt or "orc_" in txt or "oracle/" in txt:
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_python_planner_cuts_where_the_library_planner_cuts():
    """mpc_amd.sharding.split_range is plain Python; mpcx_plan_split is what the C++ engines and the group entry points
    use.  Both must cut a vector in the same places (a host function of the library: no GPU needed)."""
    from mpc_amd.lib import load_library
    from mpc_amd.sharding import split_range
    lib = load_library()
    first, count = ctypes.c_uint64(), ctypes.c_uint64()
    for total in (0, 1, 63, 64, 65, 127, 4096, 12345, (1 << 24) + 77, 1 << 28):
        for parts in (1, 2, 3, 7, 8, 16):
            for granule in (1, 64, 128, 1000):
                for part in range(parts):
                    rc = lib.mpcx_plan_split(ctypes.c_uint64(total), ctypes.c_int(parts), ctypes.c_int(part),
                                             ctypes.c_uint64(granule), ctypes.byref(first), ctypes.byref(count))
                    assert rc == 0
                    assert split_range(total, parts, part, granule) == (first.value, count.value)
    assert lib.mpcx_plan_split(ctypes.c_uint64(8), ctypes.c_int(2), ctypes.c_int(2), ctypes.c_uint64(1),
                               ctypes.byref(first), ctypes.byref(count)) != 0
    with pytest.raises(ValueError):
        split_range(8, 2, 2)
