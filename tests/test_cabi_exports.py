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
