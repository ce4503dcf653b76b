"""GPU parity tests of the MyEngine path (run with -m gpu on an MI355X): every HIP segment, called through the
C ABI, against the CPU oracle, message by message, plus the identities of test/test_Myengine.cpp."""
import json
import os

import numpy as np
import pytest

import my_protocols as mp
import protocols as pr
from oracle_lib import KEY01, KEY12, KEY20, KEYS, cks, prg_pops
from test_oracle_my import _pc_case

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_goldens.json")))


@pytest.fixture(scope="module")
def gpu():
    from my_backends import GpuMy
    return GpuMy()


@pytest.fixture(scope="module")
def orm():
    from my_backends import OracleMy
    return OracleMy()


def same_logs(w_gpu, w_orc):
    assert [t for t, _ in w_gpu.log] == [t for t, _ in w_orc.log]
    for (tag, a), (_, b) in zip(w_gpu.log, w_orc.log):
        assert a.dtype == b.dtype and np.array_equal(a, b), f"message {tag} differs"


def rnd(rng, n):
    return rng.integers(-2 ** 63, 2 ** 63, int(n), dtype=np.int64)


@pytest.mark.parametrize("r,l,c,layout", [(3, 4, 5, 0), (9, 11, 10, 1), (128, 784, 128, 0), (200, 200, 200, 0), (130, 257, 66, 1),
                                           (3, 0, 5, 0), (0, 4, 5, 1), (3, 4, 0, 0)])  # empty contraction / result
def test_my_matmul_parity(gpu, orm, r, l, c, layout):
    """Includes shapes whose buffer offsets lie far beyond BUFF_NUM (extended-buffer semantics)."""
    rng = np.random.default_rng(r * 31 + c)
    Xs = [rnd(rng, r * l), rnd(rng, r * l)]
    Ys = [rnd(rng, l * c), rnd(rng, l * c)]
    wg, wo = pr.World(gpu), pr.World(orm)
    Zg = mp.my_matmul(wg, Xs, Ys, r, l, c, layout)
    Zo = mp.my_matmul(wo, Xs, Ys, r, l, c, layout)
    same_logs(wg, wo)
    for j in range(2):
        assert np.array_equal(Zg[j], Zo[j])
    tr = (lambda m, a, b: np.ascontiguousarray(m.reshape(b, a).T).ravel()) if layout else (lambda m, a, b: m)
    X = tr(orm.ring_add(Xs[0], Xs[1]), r, l)
    Y = tr(orm.ring_add(Ys[0], Ys[1]), l, c)
    assert np.array_equal(tr(mp.my_reveal(wg, Zg), r, c), orm.ring_gemm(X, Y, r, l, c))


@pytest.mark.parametrize("n", [1, 10, 4099, (1 << 20) + 5])
def test_my_vecmul_parity(gpu, orm, n):
    rng = np.random.default_rng(n)
    Xs, Ys = [rnd(rng, n), rnd(rng, n)], [rnd(rng, n), rnd(rng, n)]
    wg, wo = pr.World(gpu), pr.World(orm)
    Zg, Zo = mp.my_vecmul(wg, Xs, Ys), mp.my_vecmul(wo, Xs, Ys)
    same_logs(wg, wo)
    for j in range(2):
        assert np.array_equal(Zg[j], Zo[j])
    x = orm.ring_add(Xs[0], Xs[1]).view(np.uint64)
    y = orm.ring_add(Ys[0], Ys[1]).view(np.uint64)
    assert np.array_equal(mp.my_reveal(wg, Zg).view(np.uint64), x * y)


def test_my_vecmul_sharded_ranges(gpu, orm):
    """(n_total, i0, n) slices reproduce the whole-vector call."""
    n = 1000
    rng = np.random.default_rng(3)
    X, Y = rnd(rng, n), rnd(rng, n)
    C0, C1 = orm.my_beaver_dealer_vecmul(KEY20, KEY12, KEYS, n, 0, n)
    ef = orm.my_beaver_party_open_vecmul(X, Y, KEY20, n, 0)
    pc0, pc1, pef = [], [], []
    for i0, m in ((0, 300), (300, 1), (301, 699)):
        a, b = gpu.my_beaver_dealer_vecmul(KEY20, KEY12, KEYS, n, i0, m)
        pc0.append(a), pc1.append(b)
        pef.append(gpu.my_beaver_party_open_vecmul(X[i0:i0 + m], Y[i0:i0 + m], KEY20, n, i0))
    assert np.array_equal(np.concatenate(pc0), C0) and np.array_equal(np.concatenate(pc1), C1)
    assert np.array_equal(np.concatenate([p[: p.size // 2] for p in pef]), ef[:n])
    assert np.array_equal(np.concatenate([p[p.size // 2:] for p in pef]), ef[n:])


@pytest.mark.parametrize("n", [1, 10, 333, 4099])
def test_my_private_compare_parity(gpu, orm, n):
    x, r, beta, s0, s1 = _pc_case(n, 92)
    wg, wo = pr.World(gpu), pr.World(orm)
    bg = mp.my_private_compare(wg, [s0, s1], beta, r)
    bo = mp.my_private_compare(wo, [s0, s1], beta, r)
    same_logs(wg, wo)
    assert np.array_equal(bg, bo)
    assert np.array_equal(beta ^ bg, (x > r.view(np.uint64)).astype(np.int8))


def test_my_private_compare_arbitrary_inputs(gpu, orm):
    """Values outside the promised ranges (x shares above 66, beta outside {0,1}, r = -1) follow the reference's
    uint16 truncation and its beta == 0 test."""
    n = 257
    rng = np.random.default_rng(4)
    xb = rng.integers(0, 65536, 64 * n, dtype=np.uint16)
    beta = rng.integers(-2, 3, n, dtype=np.int8)
    r = rnd(rng, n)
    r[::7] = -1
    for party in (0, 1):
        assert np.array_equal(gpu.my_pc_party(party, xb, beta, r, 5, KEY01), orm.my_pc_party(party, xb, beta, r, 5, KEY01))


@pytest.mark.parametrize("n", [1, 10, 2000, (1 << 20) + 77])
def test_my_relu_chain_parity(gpu, orm, n):
    """share -> msb -> bit_inject, the composition of benchmark/benchmark1.cpp:195-199 (n = 2000 there)."""
    X = prg_pops(93, 0, n) // 2
    if n >= 5:
        X[:5] = [0, -1, -2, -(1 << 62), (1 << 62) - 1]
    res = []
    for be in (gpu, orm):
        w = pr.World(be)
        Xs = mp.my_share(w, X, n)
        A = mp.my_msb(w, Xs)
        Z = mp.my_bit_inject(w, Xs, A)
        res.append((w, A, Z))
    same_logs(res[0][0], res[1][0])
    for j in range(2):
        assert np.array_equal(res[0][1][j], res[1][1][j])
        assert np.array_equal(res[0][2][j], res[1][2][j])
    w = res[0][0]
    assert np.array_equal(mp.my_reveal_bools(w, res[0][1]), (X < 0).astype(np.int8))
    assert np.array_equal(mp.my_reveal(w, res[0][2]), X * (X < 0))
    if n == 2000:
        g = GOLD["my_relu_2000"]
        assert [cks(res[0][2][j]) for j in range(2)] == g["Z_share"]
        assert [[t, cks(m)] for t, m in w.log] == g["log"]


def test_my_matmul_config_golden(gpu, orm):
    n = 64
    wd = prg_pops(20, 0, 2 * n * n)
    X, Y = wd[: n * n].copy(), wd[n * n:].copy()
    w = pr.World(gpu)
    Zs = mp.my_matmul(w, mp.my_share(w, X, n * n), mp.my_share(w, Y, n * n), n, n, n)
    g = GOLD["my_matmul_64"]
    assert [cks(Zs[j]) for j in range(2)] == g["Z_share"]
    assert cks(mp.my_reveal(w, Zs)) == g["Z_plain"]
    assert [[t, cks(m)] for t, m in w.log] == g["log"]


def test_my_errors_and_device_pool(gpu):
    """Bad arguments come back as errors with a message (no kernel is launched); freed device blocks are recycled
    and a pointer the context does not own is refused."""
    import ctypes

    from mpc_amd import MpcxError
    one = np.zeros(1, dtype=np.int64)
    with pytest.raises(MpcxError, match="party"):
        gpu._go("my_beaver_party_finish_vecmul", 2, KEY20, one, np.zeros(2, np.int64), np.zeros(2, np.int64), 1, 0, 1,
                ("out", np.int64, 1))
    with pytest.raises(MpcxError, match="range outside"):
        gpu._go("my_beaver_dealer_vecmul", KEY20, KEY12, KEYS, 4, 3, 2, ("out", np.int64, 2), ("out", np.int64, 2))
    with pytest.raises(MpcxError, match="outside 2 rows"):  # rows [1, 3) of a product with two rows
        gpu._go("my_beaver_dealer_matmul_rows", KEY20, KEY12, KEYS, 2, 1, 1, 1, 2, ("out", np.int64, 2), ("out", np.int64, 2))
    # a product without rows is legal (Eigen multiplies empty matrices): nothing is written
    gpu._go("my_beaver_dealer_matmul", KEY20, KEY12, KEYS, 0, 1, 1, ("out", np.int64, 1), ("out", np.int64, 1))
    ctx = gpu.ctx
    a = ctx.empty(np.int64, 1000)
    p = a.ptr
    a.free()
    b = ctx.empty(np.int64, 1000)  # same size class: the block just freed
    assert b.ptr == p
    with pytest.raises(MpcxError, match="not allocated"):
        ctx._call("mpcx_free", ctypes.c_void_p(p + 64))
