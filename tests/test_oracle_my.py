"""CPU checks of the MyEngine restatement in oracle/ against the properties the reference's own test asserts
(test/test_Myengine.cpp): reveal(op(share(x))) == op(x) for vector_mul / matrix_mul (:45-86), privateCompare's
beta ^ beta' == (x > r) on the corner values of :101-109, msb and bit_inject on the corner values of :163-184.
The reference has no bit-pattern goldens for this engine; share bit patterns are pinned by the PRG (checked
against the real reference PRgenerator elsewhere) and by these identities."""
import numpy as np
import pytest

import my_protocols as mp
import protocols as pr
from oracle_lib import prg_pops

L_1 = (1 << 64) - 1


@pytest.fixture(scope="module")
def be():
    from my_backends import OracleMy
    return OracleMy()


def test_my_share_reveal_and_products(be):
    w = pr.World(be)
    words = prg_pops(91, 0, 4000)
    X, Y = words[:10].copy(), words[10:20].copy()
    assert np.array_equal(mp.my_reveal(w, mp.my_share(w, X, 10)), X)
    Z = mp.my_vecmul(w, mp.my_share(w, X, 10), mp.my_share(w, Y, 10))
    assert np.array_equal(mp.my_reveal(w, Z), (X.view(np.uint64) * Y.view(np.uint64)).view(np.int64))
    r, l, c = 9, 11, 10
    A, B = words[100:100 + r * l].copy(), words[300:300 + l * c].copy()
    for layout in (0, 1):
        tr = (lambda m, a, b: np.ascontiguousarray(m.reshape(a, b).T).ravel()) if layout else (lambda m, a, b: m)
        As, Bs = mp.my_share(w, A, r * l), mp.my_share(w, B, l * c)
        As, Bs = [tr(s, r, l) for s in As], [tr(s, l, c) for s in Bs]
        Zs = mp.my_matmul(w, As, Bs, r, l, c, layout)
        got = mp.my_reveal(w, Zs)
        want = be.ring_gemm(A, B, r, l, c)
        assert np.array_equal(got, tr(want, r, c))


def _pc_case(n, seed):
    w = prg_pops(seed, 0, 67 * n).reshape(n, 67)
    x = (w[:, 0].view(np.uint64) % np.uint64(L_1)).astype(np.uint64)
    r = (w[:, 1].view(np.uint64) % np.uint64(L_1)).astype(np.uint64)
    if n >= 8:  # test/test_Myengine.cpp:101-109
        x[0], x[1], r[3], r[4] = 0, L_1 - 1, 0, L_1 - 1
        x[5] = r[5]
        x[6] = (int(r[6]) + 1) % L_1
        x[7] = (int(r[7]) - 1) % L_1
    beta = (w[:, 2].view(np.uint64) % np.uint64(2)).astype(np.int8)
    s0 = ((w[:, 3:].view(np.uint64) & np.uint64(0xFFFF)) % np.uint64(67)).astype(np.uint16)
    bits = ((x[:, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(np.uint16)
    s1 = ((bits + 67 - s0) % 67).astype(np.uint16)
    return x, r.view(np.int64), beta, np.ascontiguousarray(s0).ravel(), np.ascontiguousarray(s1).ravel()


@pytest.mark.parametrize("n", [10, 333])
def test_my_private_compare(be, n):
    x, r, beta, s0, s1 = _pc_case(n, 92)
    b2 = mp.my_private_compare(pr.World(be), [s0, s1], beta, r)
    assert np.array_equal(beta ^ b2, (x > r.view(np.uint64)).astype(np.int8))


@pytest.mark.parametrize("n", [10, 1000, 5000])
def test_my_msb_and_bit_inject(be, n):
    X = prg_pops(93, 0, n) // 2
    X[:5] = [0, -1, -2, -(1 << 62), (1 << 62) - 1]  # test/test_Myengine.cpp:163-167
    w = pr.World(be)
    Xs = mp.my_share(w, X, n)
    Xs[0][2] = -1  # :170: both parties overwrite their share of element 2 with -1, the value stays -2
    Xs[1][2] = -1
    A = mp.my_msb(w, Xs)
    assert np.array_equal(mp.my_reveal_bools(w, A), (X < 0).astype(np.int8))
    Z = mp.my_bit_inject(w, Xs, A)
    assert np.array_equal(mp.my_reveal(w, Z), X * (X < 0))


def test_my_goldens_reproduce(be):
    """The committed MyEngine goldens are what tests/golden/make_goldens.py produces from the oracle today."""
    import json
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_goldens
    gold = json.load(open(os.path.join(here, "golden", "oracle_goldens.json")))
    fresh = make_goldens.build(be)
    for k in ("my_matmul_64", "my_relu_2000"):
        assert fresh[k] == gold[k]
