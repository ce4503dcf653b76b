"""ctypes access to the CPU oracle (oracle/liboracle.so) and, when built, the real reference
PRgenerator (oracle/_ref/libref_prg.so).  Test infrastructure only."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE = os.path.join(ROOT, "oracle", "liboracle.so")
_REF = os.path.join(ROOT, "oracle", "_ref", "libref_prg.so")

U64 = ctypes.c_uint64
KEY01, KEY12, KEY20, KEYS = 0xAB01, 0xBC12, 0xCA20, 0xF0


def _arg(a):
    if a is None:
        return ctypes.c_void_p(0)
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"], "oracle arrays must be contiguous"
        return ctypes.c_void_p(a.ctypes.data)
    if isinstance(a, ctypes.c_void_p):
        return a
    if isinstance(a, (int, np.integer)):
        return U64(int(a) & 0xFFFFFFFFFFFFFFFF)
    raise TypeError(type(a))


class _Lib:
    def __init__(self, path):
        self.path = path
        self.lib = ctypes.CDLL(path)

    def __getattr__(self, name):
        fn = getattr(self.lib, name)
        fn.restype = ctypes.c_int

        def call(*args):
            rc = fn(*[_arg(a) for a in args])
            if rc != 0:
                raise RuntimeError(f"{name} returned {rc}")
            return rc

        return call

    def raw(self, name, restype):
        fn = getattr(self.lib, name)
        fn.restype = restype
        return lambda *args: fn(*[_arg(a) for a in args])


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(_ORACLE):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle` or __graft_entry__.build()")
        _oracle = _Lib(_ORACLE)
    return _oracle


def ref_prg():
    """The real reference PRgenerator, or None when oracle/_ref was not built."""
    global _ref
    if _ref is None and os.path.exists(_REF):
        _ref = _Lib(_REF)
    return _ref


def fnv1a64(words):
    w = np.ascontiguousarray(words).view(np.uint64).ravel()
    return oracle().raw("orc_fnv1a64_words", ctypes.c_uint64)(w, w.size)


def cks(a):
    """FNV-1a-64 of any array's bytes (zero-padded to a multiple of 8), as 16 hex digits."""
    b = np.ascontiguousarray(a).view(np.uint8).ravel()
    pad = (-b.size) % 8
    if pad:
        b = np.concatenate([b, np.zeros(pad, np.uint8)])
    return f"{fnv1a64(b.view(np.uint64)):016X}"


def i64(n):
    return np.zeros(int(n), dtype=np.int64)


def i8(n):
    return np.zeros(int(n), dtype=np.int8)


def u16(n):
    return np.zeros(int(n), dtype=np.uint16)


def prg_pops(seed, first, count, lcg=0):
    out = i64(count)
    oracle().orc_prg_fill_pop(seed, lcg, first, count, out)
    return out


def prg_buff(seed, first, count, lcg=0):
    a, b = i64(count), i8(count)
    oracle().orc_prg_fill_buff(seed, lcg, first, count, a, b)
    return a, b
