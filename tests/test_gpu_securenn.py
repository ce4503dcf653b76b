"""GPU parity tests of the SecureNN path (run with -m gpu on an MI355X): every HIP segment, called
through the C ABI, against the CPU oracle -- per segment, per protocol message, on golden checksums --
plus the algebraic properties the reference's own tests assert (test/test_securenn.cpp)."""
import json
import os

import numpy as np
import pytest

import protocols as pr
from oracle_lib import KEY01, KEYS, cks, prg_pops

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_goldens.json")))


@pytest.fixture(scope="module")
def gpu():
    from backends import GpuBackend
    return GpuBackend()


def same_logs(w_gpu, w_orc):
    assert [t for t, _ in w_gpu.log] == [t for t, _ in w_orc.log]
    for (tag, a), (_, b) in zip(w_gpu.log, w_orc.log):
        assert a.dtype == b.dtype and np.array_equal(a, b), f"message {tag} differs"


def rnd(rng, n):
    return rng.integers(-2 ** 63, 2 ** 63, int(n), dtype=np.int64)


def pc_inputs(n, seed=22):
    w = prg_pops(seed, 0, 67 * n).reshape(n, 67)
    x, r = w[:, 0].copy(), w[:, 1].copy()
    r[2::5] = -1
    beta = (w[:, 2].view(np.uint64) % 2).astype(np.int8)
    s0 = ((w[:, 3:].view(np.uint64) & 0xFFFF) % 67).astype(np.uint16)
    bits = ((x.view(np.uint64)[:, None] >> np.arange(64, dtype=np.uint64)) & 1).astype(np.uint16)
    s1 = ((bits + 67 - s0) % 67).astype(np.uint16)
    return x, r, beta, np.ascontiguousarray(s0).ravel(), np.ascontiguousarray(s1).ravel()


def trunc_div2(a):
    return (a // 2 + ((a < 0) & (a % 2 != 0))).astype(np.int64)


@pytest.mark.parametrize("n", [1, 5, 500, 4099])
def test_private_compare_parity(gpu, orc, n):
    x, r, beta, s0, s1 = pc_inputs(n)
    wg, wo = pr.World(gpu), pr.World(orc)
    bg = pr.snn_private_compare(wg, [s0, s1], beta, r)
    bo = pr.snn_private_compare(wo, [s0, s1], beta, r)
    same_logs(wg, wo)
    assert np.array_equal(bg, bo)
    assert np.array_equal(beta ^ bg, (x.view(np.uint64) > r.view(np.uint64)).astype(np.int8))
    if n == 500:
        assert [[t, cks(m)] for t, m in wg.log] == GOLD["snn_pc_500"]["log"]


def test_private_compare_sharded_ranges(gpu, orc):
    """(n_total, i0, n) slices must reproduce the whole-vector call (element-range sharding)."""
    n = 1000
    x, r, beta, s0, s1 = pc_inputs(n)
    whole = orc.snn_pc_party(1, s1, beta, r, n, 0, KEY01)
    parts = []
    for i0, m in ((0, 300), (300, 1), (301, 699)):
        parts.append(gpu.snn_pc_party(1, s1[64 * i0:64 * (i0 + m)], beta[i0:i0 + m], r[i0:i0 + m], n, i0, KEY01))
    assert np.array_equal(np.concatenate(parts), whole)


def test_private_compare_arbitrary_uint16_and_beta(gpu, orc):
    """Inputs outside [0,66] and beta outside {0,1} follow the reference's uint16 truncation rules."""
    n = 257
    rng = np.random.default_rng(4)
    xb = rng.integers(0, 65536, 64 * n, dtype=np.uint16)
    beta = rng.integers(-2, 3, n, dtype=np.int8)
    r = rnd(rng, n)
    r[::7] = -1
    for party in (0, 1):
        assert np.array_equal(gpu.snn_pc_party(party, xb, beta, r, n, 0, KEY01),
                              orc.snn_pc_party(party, xb, beta, r, n, 0, KEY01))
    d1 = rng.integers(0, 65536, 64 * n, dtype=np.uint16)
    assert np.array_equal(gpu.snn_pc_helper(xb, d1), orc.snn_pc_helper(xb, d1))


@pytest.mark.parametrize("n", [1, 63, 4099, 70001])
def test_private_compare_without_the_byte_cache(gpu, orc, n):
    """With the cache of generator bytes switched off the u bytes are generated for the elements that read them
    only (beta = 1 and r = -1, or a beta outside {0, 1}); the result is the oracle's, in element ranges too."""
    rng = np.random.default_rng(n)
    xb = rng.integers(0, 67, 64 * n, dtype=np.uint16)
    beta = rng.integers(0, 2, n, dtype=np.int8)
    beta[5::11] = 3
    r = rnd(rng, n)
    r[2::5] = -1
    want = [orc.snn_pc_party(party, xb, beta, r, n, 0, KEY01) for party in (0, 1)]
    gpu.ctx.call("set_rand_cache", 0)
    try:
        for party in (0, 1):
            assert np.array_equal(gpu.snn_pc_party(party, xb, beta, r, n, 0, KEY01), want[party])
        i0 = n // 3
        m = n - i0
        got = gpu.snn_pc_party(1, xb[64 * i0:], beta[i0:], r[i0:], n, i0, KEY01)
        assert np.array_equal(got, want[1][64 * i0:])
        none = np.zeros(n, dtype=np.int8)  # no element reads u: nothing is listed
        assert np.array_equal(gpu.snn_pc_party(0, xb, none, r, n, 0, KEY01), orc.snn_pc_party(0, xb, none, r, n, 0, KEY01))
        assert m > 0
    finally:
        gpu.ctx.call("set_rand_cache", 4 << 30)


def test_snn_matmul_config_golden(gpu, orc):
    n = 64
    wd = prg_pops(20, 0, 2 * n * n)
    X, Y = wd[: n * n].copy(), wd[n * n:].copy()
    wg, wo = pr.World(gpu), pr.World(orc)
    Zg = pr.snn_matmul(wg, pr.snn_share(wg, X, n * n), pr.snn_share(wg, Y, n * n), n, n, n)
    Zo = pr.snn_matmul(wo, pr.snn_share(wo, X, n * n), pr.snn_share(wo, Y, n * n), n, n, n)
    same_logs(wg, wo)
    g = GOLD["snn_matmul_64"]
    assert [cks(Zg[j]) for j in range(2)] == g["Z_share"] == [cks(Zo[j]) for j in range(2)]
    assert cks(pr.snn_reveal(wg, Zg)) == g["Z_plain"]


@pytest.mark.parametrize("r,l,c,layout", [(3, 4, 5, 0), (3, 4, 5, 1), (128, 784, 128, 0), (128, 128, 10, 1), (200, 200, 200, 0),
                                           (3, 0, 5, 0), (0, 4, 5, 1), (3, 4, 0, 0)])  # empty contraction / result
def test_snn_matmul_protocol_parity(gpu, orc, r, l, c, layout):
    """Includes the MLP layer shapes (beyond BUFF_NUM: extended-buffer semantics) and 200x200 of the benchmark."""
    rng = np.random.default_rng(r * 7 + c)
    Xs = [rnd(rng, r * l), rnd(rng, r * l)]
    Ys = [rnd(rng, l * c), rnd(rng, l * c)]
    wg, wo = pr.World(gpu), pr.World(orc)
    Zg = pr.snn_matmul(wg, Xs, Ys, r, l, c, layout)
    Zo = pr.snn_matmul(wo, Xs, Ys, r, l, c, layout)
    same_logs(wg, wo)
    for j in range(2):
        assert np.array_equal(Zg[j], Zo[j])
    tr = (lambda m, a, b: np.ascontiguousarray(m.reshape(b, a).T).ravel()) if layout else (lambda m, a, b: m)
    X = tr(orc.ring_add(Xs[0], Xs[1]), r, l)
    Y = tr(orc.ring_add(Ys[0], Ys[1]), l, c)
    assert np.array_equal(tr(pr.snn_reveal(wg, Zg), r, c), orc.ring_gemm(X, Y, r, l, c))


@pytest.mark.parametrize("fused", [True, False])  # parties 0 / 1 with one call per messenger gap, or segment by segment
@pytest.mark.parametrize("n", [1, 10, 1000, 2000, (1 << 20) + 77])  # the last: every kernel grid-strides
def test_snn_msb_parity(gpu, orc, n, fused, monkeypatch):
    """n = 2000 is the reference benchmark's size, which the reference itself cannot run (buffer overrun)."""
    monkeypatch.setattr(pr, "SNN_FUSED", fused)
    a = trunc_div2(prg_pops(22, 67 * 1000, n))
    res = []
    for be in (gpu, orc):
        w = pr.World(be)
        As = pr.snn_share(w, a, n)
        res.append((w, pr.snn_msb(w, As[:2])))
    same_logs(res[0][0], res[1][0])
    for j in range(2):
        assert np.array_equal(res[0][1][j], res[1][1][j])
    assert np.array_equal(pr.snn_reveal(res[0][0], res[0][1]), (a < 0).astype(np.int64))
    if n == 1000:
        g = GOLD["snn_msb_1000"]
        assert [cks(res[0][1][j]) for j in range(2)] == g["B_share"]
        assert [[t, cks(m)] for t, m in res[0][0].log] == g["log"]


def test_share_convert_corner_values(gpu, orc):
    n = 10
    A = prg_pops(KEY01, 1000, n)
    A[A == -1] = 1
    A[0], A[1], A[2], A[3] = 0, -2 ** 63, -2, 1
    A[4] = np.int64(2 ** 63 - 1)
    wg, wo = pr.World(gpu), pr.World(orc)
    Yg = pr.snn_share_convert(wg, pr.snn_share(wg, A, n)[:2])
    Yo = pr.snn_share_convert(wo, pr.snn_share(wo, A, n)[:2])
    same_logs(wg, wo)
    assert np.array_equal(gpu.snn_add_L1(Yg[0], Yg[1]), A)
    Bg = pr.snn_msb_L1(wg, Yg)
    Bo = pr.snn_msb_L1(wo, Yo)
    same_logs(wg, wo)
    assert np.array_equal(pr.snn_reveal(wg, Bg), (A < 0).astype(np.int64))
    assert np.array_equal(Bg[0], Bo[0]) and np.array_equal(Bg[1], Bo[1])


def test_snn_relu_layer(gpu, orc):
    """One SecureNN layer as the reference benchmark composes it (benchmark/benchmark1.cpp:104-136):
    matrix_mul -> local >> 16 on each share -> msb -> vector_mul(a, msb) -> a - that."""
    r, l, c = 16, 24, 8
    rng = np.random.default_rng(11)
    X = (rng.integers(-2 ** 63, 2 ** 63, r * l, dtype=np.int64) >> 48)
    W = (rng.integers(-2 ** 63, 2 ** 63, l * c, dtype=np.int64) >> 48)
    outs = []
    for be in (gpu, orc):
        w = pr.World(be)
        Zs = pr.snn_matmul(w, pr.snn_share(w, X, r * l), pr.snn_share(w, W, l * c), r, l, c)
        a = [be.ring_sar(Zs[0], 16), be.ring_sar(Zs[1], 16)]
        m = pr.snn_msb(w, a)
        am = pr.snn_vecmul(w, a, m)
        relu = [be.ring_sub(a[j], am[j]) for j in range(2)]
        outs.append((w, relu, a))
    same_logs(outs[0][0], outs[1][0])
    for j in range(2):
        assert np.array_equal(outs[0][1][j], outs[1][1][j])
    got = pr.snn_reveal(outs[0][0], outs[0][1])
    a_plain = orc.ring_add(outs[1][2][0], outs[1][2][1])
    assert np.array_equal(got, np.maximum(a_plain, 0))


def test_snn_mlp_config4(gpu, orc):
    """The 784-128-128-10 MLP at batch 128: input and W1..W3 are PRgenerator(24,0).pop_int64() >> 48 in that order;
    per layer matrix_mul -> local >> 16 -> msb -> vector_mul(a, msb) -> a - that, no ReLU after the last layer
    (the composition of benchmark/benchmark1.cpp:104-136).  The Beaver triples of the first layer alone need
    417 792 buffer words, far beyond BUFF_NUM: every message and share must still equal the oracle's."""
    dims = [(128, 784, 128), (128, 128, 128), (128, 128, 10)]
    words = prg_pops(24, 0, 128 * 784 + sum(l * c for _, l, c in dims)) >> 48
    X, off = words[: 128 * 784].copy(), 128 * 784
    Ws = []
    for _, l, c in dims:
        Ws.append(words[off: off + l * c].copy())
        off += l * c
    outs = []
    for be in (gpu, orc):
        w = pr.World(be)
        act = pr.snn_share(w, X, X.size)[:2]
        for li, (r, l, c) in enumerate(dims):
            Wsh = pr.snn_share(w, Ws[li], l * c)[:2]
            Z = pr.snn_matmul(w, act, Wsh, r, l, c)
            a = [be.ring_sar(Z[0], 16), be.ring_sar(Z[1], 16)]
            if li < 2:
                m = pr.snn_msb(w, a)
                am = pr.snn_vecmul(w, a, m)
                a = [be.ring_sub(a[j], am[j]) for j in range(2)]
            act = a
        outs.append((w, act))
    same_logs(outs[0][0], outs[1][0])
    for j in range(2):
        assert np.array_equal(outs[0][1][j], outs[1][1][j])
    # plaintext network with the same fixed-point rule; local truncation of shares is off by at most one per layer
    h = X.reshape(128, 784)
    for li, (r, l, c) in enumerate(dims):
        h = (h @ Ws[li].reshape(l, c)) >> 16
        if li < 2:
            h = np.maximum(h, 0)
    got = pr.snn_reveal(outs[0][0], outs[0][1]).reshape(128, 10)
    assert np.abs(got - h).max() <= 64


def test_snn_small_segments_parity(gpu, orc):
    n = 333
    rng = np.random.default_rng(8)
    a, b = rnd(rng, n), rnd(rng, n)
    a[:3] = [-1, -1, 0]
    b[:3] = [-1, 1, -1]
    assert np.array_equal(gpu.snn_add_L1(a, b), orc.snn_add_L1(a, b))
    z0, z1 = rng.integers(0, 65536, 64 * n, dtype=np.uint16), rng.integers(0, 65536, 64 * n, dtype=np.uint16)
    assert np.array_equal(gpu.snn_add_zp(z0, z1), orc.snn_add_zp(z0, z1))
    for party in range(3):
        assert np.array_equal(gpu.snn_share_local(party, a, n, KEY01, 17), orc.snn_share_local(party, a, n, KEY01, 17))
    for got, want in zip(gpu.snn_beaver_dealer_vecmul(KEYS, 1000, 100, n), orc.snn_beaver_dealer_vecmul(KEYS, 1000, 100, n)):
        assert np.array_equal(got, want)
    for got, want in zip(gpu.snn_msbl1_helper_deal(1000, 100, n, KEYS), orc.snn_msbl1_helper_deal(1000, 100, n, KEYS)):
        assert np.array_equal(got, want)
    for got, want in zip(gpu.snn_sc_helper_split(a, b, 1000, 100, KEYS), orc.snn_sc_helper_split(a, b, 1000, 100, KEYS)):
        assert np.array_equal(got, want)
    for party in (0, 1):
        assert np.array_equal(gpu.snn_sc_party_open(party, a, 1000, 100, KEY01), orc.snn_sc_party_open(party, a, 1000, 100, KEY01))
        assert np.array_equal(gpu.snn_sc_party_finish(party, a, b, orc.snn_add_L1(a, a), 1000, 100, KEY01),
                              orc.snn_sc_party_finish(party, a, b, orc.snn_add_L1(a, a), 1000, 100, KEY01))


@pytest.mark.timeout(400)
def test_private_compare_range_beyond_32_bit_stream_offsets(gpu, orc):
    """A range at the end of a 2^24 + 2^23 element vector: its u words sit past stream word 2^32 of the s01
    generator (256 stream words per element), which is where an 8-GPU run of 2^24 elements per GPU works."""
    nt = (1 << 24) + (1 << 23)
    n, i0 = 1003, nt - 1003
    x, r, beta, s0, s1 = pc_inputs(n)
    beta[:] = 1  # with r = -1 on every fifth element: the branch that reads u (SecurennEngine.cpp:577-585)
    want = orc.snn_pc_party(0, s0, beta, r, nt, i0, KEY01)
    got = gpu.snn_pc_party(0, s0, beta, r, nt, i0, KEY01)
    assert np.array_equal(got, want)
    ints_o, bools_o = orc.prg_fill_buff(KEY01, 128 * nt - 700, 700)
    ints_g, bools_g = gpu.prg_fill_buff(KEY01, 128 * nt - 700, 700)
    assert np.array_equal(ints_g, ints_o) and np.array_equal(bools_g, bools_o)


def test_l1_addition_corner_values(gpu, orc):
    """add_L_1 (SecurennEngine.cpp:15-21) on every pair of 0, 1, 2^63 - 1, 2^63, 2^64 - 2, 2^64 - 1 and random words."""
    corner = np.array([0, 1, (1 << 63) - 1, 1 << 63, (1 << 64) - 2, (1 << 64) - 1], dtype=np.uint64)
    rng = np.random.default_rng(9)
    a = np.concatenate([np.repeat(corner, 6), rng.integers(0, 1 << 64, 1000, dtype=np.uint64)]).view(np.int64)
    b = np.concatenate([np.tile(corner, 6), rng.integers(0, 1 << 64, 1000, dtype=np.uint64)]).view(np.int64)
    assert np.array_equal(gpu.snn_add_L1(a, b), orc.snn_add_L1(a, b))
