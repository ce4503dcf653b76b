"""GPU parity tests (run with -m gpu on an MI355X): every HIP segment of the ABY3 path, called through
the C ABI, must reproduce the CPU oracle bit for bit -- per segment, per protocol message and on the
committed golden checksums."""
import json
import os

import numpy as np
import pytest

import protocols as pr
from oracle_lib import KEY01, KEY12, KEY20, KEYS, cks, prg_pops

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


@pytest.mark.parametrize("seed", [KEYS, KEY01, KEY12, KEY20, 20])
@pytest.mark.parametrize("first,count", [(0, 1), (0, 257), (255, 2), (256, 256), (1000, 70001), (65535, 131073)])
def test_prg_pop_parity(gpu, orc, seed, first, count):
    assert np.array_equal(gpu.prg_fill_pop(seed, first, count), orc.prg_fill_pop(seed, first, count))


@pytest.mark.parametrize("first,count", [(0, 1), (0, 160000), (127, 130), (159990, 4000), (1 << 20, 100003)])
def test_prg_buff_parity(gpu, orc, first, count):
    gi, gb = gpu.prg_fill_buff(KEY01, first, count)
    oi, ob = orc.prg_fill_buff(KEY01, first, count)
    assert np.array_equal(gi, oi) and np.array_equal(gb, ob)


def test_prg_nonzero_lcg(gpu, orc):
    assert np.array_equal(gpu.prg_fill_pop(0x1234567, 3, 5000, lcg=99), orc.prg_fill_pop(0x1234567, 3, 5000, lcg=99))


@pytest.mark.parametrize("r,l,c", [(1, 1, 1), (5, 4, 3), (64, 64, 64), (128, 128, 128), (129, 130, 131), (200, 200, 200),
                                   (3, 1000, 7), (300, 17, 260), (128, 784, 128), (128, 128, 10),
                                   (512, 256, 512), (600, 130, 520), (1024, 64, 768)])
@pytest.mark.parametrize("lay", [0, 1])
def test_ring_gemm_parity(gpu, orc, r, l, c, lay):
    rng = np.random.default_rng(r * 1000003 + l * 1009 + c)
    A, B = rnd(rng, r * l), rnd(rng, l * c)
    want = orc.ring_gemm(A, B, r, l, c, lay, lay, lay)
    assert np.array_equal(gpu.ring_gemm(A, B, r, l, c, lay, lay, lay), want)


def test_ring_gemm_extreme_digits(gpu, orc):
    """Operands whose signed digits are all -128 / +127 (largest accumulator magnitudes) and K chunking."""
    r, l, c = 130, 9000, 129  # l > 8192 forces two accumulation chunks
    for va, vb in ((0x8080808080808080, 0x8080808080808080), (0x7F7F7F7F7F7F7F7F, 0x8080808080808080),
                   (0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF)):
        A = np.full(r * l, va, dtype=np.uint64).view(np.int64)
        B = np.full(l * c, vb, dtype=np.uint64).view(np.int64)
        assert np.array_equal(gpu.ring_gemm(A, B, r, l, c), orc.ring_gemm(A, B, r, l, c))


def test_ring_gemm_accumulate_and_mixed_layouts(gpu, orc):
    r, l, c = 70, 90, 50
    rng = np.random.default_rng(5)
    A, B, C0 = rnd(rng, r * l), rnd(rng, l * c), rnd(rng, r * c)
    for la, lb, lc in ((0, 1, 0), (1, 0, 1), (1, 1, 0)):
        assert np.array_equal(gpu.ring_gemm(A, B, r, l, c, la, lb, lc, C=C0), orc.ring_gemm(A, B, r, l, c, la, lb, lc, C=C0))


def test_ring_gemm_random_shapes_and_layouts(gpu, orc):
    """Forty random shapes (with every dimension equal to 1 among them) in every combination of operand and result
    layouts, with and without accumulation, against numpy's wrapping uint64 product."""
    rng = np.random.default_rng(2024)
    shapes = [(1, 1, 7), (1, 9, 1), (6, 1, 1), (1, 70, 130), (130, 70, 1), (65, 1, 67)]
    shapes += [tuple(int(v) for v in rng.integers(1, 300, 3)) for _ in range(34)]
    for k, (r, l, c) in enumerate(shapes):
        la, lb, lc, acc = k & 1, (k >> 1) & 1, (k >> 2) & 1, (k >> 3) & 1
        A = rng.integers(0, 1 << 64, (r, l), dtype=np.uint64)
        B = rng.integers(0, 1 << 64, (l, c), dtype=np.uint64)
        C0 = rng.integers(0, 1 << 64, (r, c), dtype=np.uint64)
        flat = lambda M, lay: (np.ascontiguousarray(M.T) if lay else np.ascontiguousarray(M)).ravel().view(np.int64)
        with np.errstate(over="ignore"):
            want = A @ B + (C0 if acc else np.uint64(0))
        got = gpu.ring_gemm(flat(A, la), flat(B, lb), r, l, c, la, lb, lc, C=flat(C0, lc) if acc else None)
        got = got.view(np.uint64)
        got = got.reshape(c, r).T if lc else got.reshape(r, c)
        assert np.array_equal(got, want), f"shape {(r, l, c)} layouts {(la, lb, lc)} accumulate {acc}"


@pytest.mark.parametrize("mode", ["0", "2"])
def test_ring_gemm_both_formulations(gpu, orc, mode, monkeypatch):
    """The positional (36 digit-pair products) and the residue (18 products + CRT) kernels, each forced for every
    shape: random shapes and layouts, extreme digits with two K chunks, accumulation, and the tile-ragged large case."""
    monkeypatch.setenv("MPCX_GEMM_RNS", mode)
    test_ring_gemm_random_shapes_and_layouts(gpu, orc)
    test_ring_gemm_extreme_digits(gpu, orc)
    test_ring_gemm_accumulate_and_mixed_layouts(gpu, orc)
    for r, l, c, lay in ((129, 130, 131, 0), (300, 17, 260, 1), (600, 130, 520, 0), (257, 97, 255, 1)):
        test_ring_gemm_parity(gpu, orc, r, l, c, lay)
    # the largest integer products one accumulation chunk can hold: K = 8192 words whose eight digits are all -128
    # (the integer -0x8080808080808080) against themselves and against the all-(+127) word
    r, l, c = 40, 8192, 33
    lo, hi = 0x7F7F7F7F7F7F7F80, 0x7F7F7F7F7F7F7F7F
    for va, vb in ((lo, lo), (lo, hi), (hi, hi)):
        A = np.full(r * l, va, dtype=np.uint64).view(np.int64)
        B = np.full(l * c, vb, dtype=np.uint64).view(np.int64)
        assert np.array_equal(gpu.ring_gemm(A, B, r, l, c), orc.ring_gemm(A, B, r, l, c))


@pytest.mark.parametrize("tile", ["0", "1"])
def test_rns_gemm_block_tile_shapes(gpu, orc, tile, monkeypatch):
    """The residue kernel's two block tiles (256 x 256; 384 x 256 with a last tile row of 128, 256 or 384 rows), forced
    for small products whose row counts hit every case, K from one stage to several, against the oracle word for word."""
    monkeypatch.setenv("MPCX_GEMM_RNS", "2")
    monkeypatch.setenv("MPCX_RNS_TILE", tile)
    rng = np.random.default_rng(5)
    for r, l, c in ((100, 40, 70), (200, 64, 256), (384, 65, 300), (500, 130, 257), (650, 200, 64), (800, 333, 520),
                    (1153, 96, 129)):
        A, B = rnd(rng, r * l), rnd(rng, l * c)
        assert np.array_equal(gpu.ring_gemm(A, B, r, l, c), orc.ring_gemm(A, B, r, l, c)), (tile, r, l, c)


@pytest.mark.parametrize("knob,value", [("MPCX_GEMM_GM", "0"), ("MPCX_GEMM_GM", "-3"), ("MPCX_GEMM_GM", "1000000"),
                                        ("MPCX_RNS_GM", "0"), ("MPCX_RNS_GM", "99"), ("MPCX_GEMM_ABLATE", "1"),
                                        ("MPCX_GEMM_ABLATE", "3")])
def test_ring_gemm_ignores_bad_tuning_knobs(gpu, orc, knob, value, monkeypatch):
    """Environment variables cannot make the product wrong: the tile-group knobs are clamped to [1, tile rows] and the
    timing ablations are not compiled into the library."""
    monkeypatch.setenv(knob, value)
    r, l, c = 300, 130, 260
    rng = np.random.default_rng(11)
    A, B = rnd(rng, r * l), rnd(rng, l * c)
    want = orc.ring_gemm(A, B, r, l, c)
    for mode in ("0", "2"):
        monkeypatch.setenv("MPCX_GEMM_RNS", mode)
        assert np.array_equal(gpu.ring_gemm(A, B, r, l, c), want), (knob, value, mode)


def test_relayout(gpu):
    a = np.arange(37 * 91, dtype=np.int64)
    assert np.array_equal(gpu.relayout(a, 37, 91, 0, 1), np.ascontiguousarray(a.reshape(37, 91).T).ravel())
    assert np.array_equal(gpu.relayout(a, 37, 91, 1, 0), np.ascontiguousarray(a.reshape(91, 37).T).ravel())


def test_aby3_matmul_config1_golden(gpu, orc):
    """config 1 (64x64 from PRgenerator(20,0)): shares, messages and revealed product."""
    n = 64
    wd = prg_pops(20, 0, 2 * n * n)
    X, Y = wd[: n * n].copy(), wd[n * n:].copy()
    wg, wo = pr.World(gpu), pr.World(orc)
    Zg = pr.aby3_matmul(wg, pr.aby3_share(wg, X, n * n), pr.aby3_share(wg, Y, n * n), n, n, n)
    Zo = pr.aby3_matmul(wo, pr.aby3_share(wo, X, n * n), pr.aby3_share(wo, Y, n * n), n, n, n)
    same_logs(wg, wo)
    g = GOLD["aby3_matmul_64"]
    assert [cks(Zg[j][0]) for j in range(3)] == g["Z_share0"]
    assert cks(pr.aby3_reveal(wg, Zg)[0]) == g["Z_plain"]
    assert [[t, cks(m)] for t, m in wg.log] == g["log"]


@pytest.mark.parametrize("r,l,c,layout", [(5, 4, 3, 0), (5, 4, 3, 1), (200, 200, 200, 0), (130, 257, 66, 1),
                                           (520, 140, 515, 0), (512, 512, 512, 1),
                                           (5, 0, 3, 0), (5, 0, 3, 1), (0, 4, 3, 0), (5, 4, 0, 1)])  # empty contraction / result
def test_aby3_matmul_protocol_parity(gpu, orc, r, l, c, layout):
    rng = np.random.default_rng(r + l + c)
    X, Y = rnd(rng, r * l), rnd(rng, l * c)
    out = []
    for be in (gpu, orc):
        w = pr.World(be)
        Xs, Ys = pr.aby3_share(w, X, r * l), pr.aby3_share(w, Y, l * c)
        if layout:
            cm = lambda S, a, b: [tuple(np.ascontiguousarray(s.reshape(a, b).T).ravel() for s in sh) for sh in S]
            Xs, Ys = cm(Xs, r, l), cm(Ys, l, c)
        Zs = pr.aby3_matmul(w, Xs, Ys, r, l, c, layout)
        out.append((w, pr.aby3_reveal(w, Zs)[0]))
    same_logs(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][1], orc.ring_gemm(X, Y, r, l, c))


def test_aby3_matmul_row_blocks_equal_whole(gpu, orc):
    """Row-block form used for multi-GPU sharding: blocks of Z0 with their mask slices == the whole call."""
    r, l, c = 96, 40, 72
    rng = np.random.default_rng(9)
    X0, X1, Y0, Y1 = rnd(rng, r * l), rnd(rng, r * l), rnd(rng, l * c), rnd(rng, l * c)
    whole = orc.aby3_matmul_local(X0, X1, Y0, Y1, r, l, c, 0, KEY01, 11, KEY20, 7)
    parts = []
    for row0, rows in ((0, 32), (32, 40), (72, 24)):
        parts.append(gpu.aby3_matmul_local_rows(X0[row0 * l:(row0 + rows) * l], X1[row0 * l:(row0 + rows) * l], Y0, Y1,
                                                rows, l, c, row0, KEY01, 11, KEY20, 7))
    assert np.array_equal(np.concatenate(parts), whole)


@pytest.mark.parametrize("mode,tile_at_use", [("2", None), ("2", "0"), ("1", None), ("0", None)])
@pytest.mark.parametrize("rows,l,c,lay_x,lay_y,row0", [(300, 2100, 1100, 0, 0, 0), (130, 700, 260, 1, 1, 5),
                                                         (64, 5000, 300, 0, 1, 17), (1030, 2048, 1024, 1, 0, 0)])
def test_aby3_matmul_with_a_prepared_second_operand(gpu, orc, rows, l, c, lay_x, lay_y, row0, mode, tile_at_use, monkeypatch):
    """mpcx_aby3_operand_prepare + mpcx_aby3_matmul_local_prepared = the ordinary segment, word for word: with the
    residue form forced (planes made once and read by two products with different row counts; l = 5000 runs as two
    accumulation chunks), with the block tile switched between preparation and use (the planes no longer fit and the
    product splits its operand itself), with the formulation chosen by the library, and with the digit-pair form
    forced (no planes are made)."""
    import ctypes
    ctx = gpu.ctx
    monkeypatch.setenv("MPCX_GEMM_RNS", mode)
    rng = np.random.default_rng(rows + l + c)
    Y0, Y1 = rnd(rng, l * c), rnd(rng, l * c)
    to_lay = lambda M, a, b, lay: np.ascontiguousarray(M.reshape(a, b).T).ravel() if lay else M
    dY0, dY1 = ctx.to_device(to_lay(Y0, l, c, lay_y)), ctx.to_device(to_lay(Y1, l, c, lay_y))
    op = ctypes.c_void_p()
    U64 = ctypes.c_uint64
    ctx._call("mpcx_aby3_operand_prepare", ctypes.c_void_p(dY0.ptr), ctypes.c_void_p(dY1.ptr), U64(l), U64(c),
              ctypes.c_int(lay_y), ctypes.byref(op))
    ctx.lib.mpcx_aby3_operand_bytes.restype = ctypes.c_uint64
    held = ctx.lib.mpcx_aby3_operand_bytes(op)
    if mode == "2":
        assert held >= 18 * 2 * l * c  # the planes of [Y0 + Y1 ; Y0]
    if mode == "0":
        assert held == 0
    if tile_at_use is not None:
        monkeypatch.setenv("MPCX_RNS_TILE", tile_at_use)
    try:
        for rr in (rows, max(1, rows // 3)):
            X0, X1 = rnd(rng, rr * l), rnd(rng, rr * l)
            want = orc.aby3_matmul_local_rows(X0, X1, Y0, Y1, rr, l, c, row0, KEY01, 11, KEY20, 7) \
                if hasattr(orc, "aby3_matmul_local_rows") else None
            dX0, dX1 = ctx.to_device(to_lay(X0, rr, l, lay_x)), ctx.to_device(to_lay(X1, rr, l, lay_x))
            Z = ctx.empty(np.int64, rr * c)
            ctx._call("mpcx_aby3_matmul_local_prepared", ctypes.c_void_p(dX0.ptr), ctypes.c_void_p(dX1.ptr),
                      ctypes.c_int(lay_x), op, U64(rr), U64(row0), U64(KEY01), U64(11), U64(KEY20), U64(7),
                      ctypes.c_void_p(Z.ptr))
            plain = gpu.aby3_matmul_local_rows(X0, X1, Y0, Y1, rr, l, c, row0, KEY01, 11, KEY20, 7)
            got = Z.to_host()
            assert np.array_equal(got, plain), "prepared product differs from the ordinary segment"
            if want is not None:
                assert np.array_equal(got, want)
            else:  # the oracle has no row-block form: the mask is pops [row0 c, ..) of both streams, the product exact
                from oracle_lib import prg_pops
                mask = prg_pops(KEY01, 11 + row0 * c, rr * c) - prg_pops(KEY20, 7 + row0 * c, rr * c)
                with np.errstate(over="ignore"):
                    assert np.array_equal(got, orc.ring_gemm(X0, Y0 + Y1, rr, l, c) + orc.ring_gemm(X1, Y0, rr, l, c) + mask)
        ctx.lib.mpcx_aby3_operand_uses.restype = ctypes.c_uint64
        uses = ctx.lib.mpcx_aby3_operand_uses(op)
        if mode == "2":
            assert uses == (2 if tile_at_use is None else 0)  # both products read the planes / neither could
        if mode == "0":
            assert uses == 0
    finally:
        ctx._call("mpcx_aby3_operand_free", op)


@pytest.mark.parametrize("fused", [True, False])  # msb as one call per messenger gap, or as separate post / pre calls
@pytest.mark.parametrize("n", [1, 4, 63, 64, 65, 1000, 4097, (1 << 20) + 4099])  # the last: every kernel grid-strides
def test_aby3_relu_chain_parity(gpu, orc, n, fused, monkeypatch):
    """share -> add_prepare -> msb -> bit_inject: every message and every share equals the oracle's."""
    monkeypatch.setattr(pr, "MSB_FUSED", fused)
    X = prg_pops(21, 0, n)
    X[: min(n, 6)] = np.array([0, -1, 1, -2 ** 63, 2 ** 63 - 1, -2], dtype=np.int64)[: min(n, 6)]
    res = []
    for be in (gpu, orc):
        w = pr.World(be)
        Xs = pr.aby3_share(w, X, n)
        Gs, Ps = pr.aby3_add_prepare(w, Xs)
        Zb = pr.aby3_msb(w, Gs, Ps)
        pops = [list(d) for d in w.pops]
        C = pr.aby3_bit_inject(w, Xs, Zb)
        # the same call again with the minuend: the GPU's post segment subtracts in the same pass
        # (mpcx_aby3_bit_inject_p*_post_sub), the oracle runs its post segment and then ring_sub
        w2 = pr.World(be, record=False)
        w2.pops = pops
        R = pr.aby3_bit_inject(w2, Xs, Zb, minuend=Xs)
        res.append((w, Zb, C, R, Xs))
    same_logs(res[0][0], res[1][0])
    for j in range(3):
        for i in range(2):
            assert np.array_equal(res[0][1][j][i], res[1][1][j][i])
            assert np.array_equal(res[0][2][j][i], res[1][2][j][i])
            assert np.array_equal(res[0][3][j][i], res[1][3][j][i])
            assert np.array_equal(res[0][3][j][i], orc.ring_sub(res[1][4][j][i], res[1][2][j][i]))
    assert np.array_equal(pr.aby3_reveal(res[0][0], res[0][3])[0], np.maximum(X, 0))
    w = res[0][0]
    assert np.array_equal(pr.aby3_reveal_bools(w, res[0][1])[0], (X < 0).astype(np.int8))
    assert np.array_equal(pr.aby3_reveal(w, res[0][2])[0], X * (X < 0))
    if n == 1000:
        g = GOLD["aby3_relu_1000"]
        assert [[t, cks(m)] for t, m in w.log] == g["log"]
        assert [cks(res[0][2][j][0]) for j in range(3)] == g["C_share0"]


def test_aby3_misc_segments_parity(gpu, orc):
    n = 777
    rng = np.random.default_rng(2)
    a, b, c = rnd(rng, n), rnd(rng, n), rnd(rng, n)
    assert np.array_equal(gpu.aby3_reveal3(a, b, c), orc.aby3_reveal3(a, b, c))
    ab, bb, cb = [(x & 1).astype(np.int8) for x in (a, b, c)]
    assert np.array_equal(gpu.aby3_reveal3_bools(ab, bb, cb), orc.aby3_reveal3_bools(ab, bb, cb))
    assert np.array_equal(gpu.aby3_vecmul_local(a, b, c, a, n, KEY01, 5, KEY20, 9),
                          orc.aby3_vecmul_local(a, b, c, a, n, KEY01, 5, KEY20, 9))
    for party in range(3):
        assert np.array_equal(gpu.aby3_trunc_local(party, a, b, 16, KEY12, 3),
                              orc.aby3_trunc_local(party, a, b, 16, KEY12, 3))
    assert np.array_equal(gpu.aby3_share_bools_local(ab, n, KEY01, 1, KEY20, 2),
                          orc.aby3_share_bools_local(ab, n, KEY01, 1, KEY20, 2))
    assert np.array_equal(gpu.ring_sar(a, 16), orc.ring_sar(a, 16))
    assert np.array_equal(gpu.ring_shl1(a), orc.ring_shl1(a))
    assert np.array_equal(gpu.ring_sub(a, b), orc.ring_sub(a, b))


@pytest.mark.parametrize("n", [1, 100, 4099, (1 << 18) + 5])
def test_aby3_vecbool_mul_and_reveal_bits_parity(gpu, orc, n):
    """vecbool_mul's local product (ABY3Engine.cpp:253-256; n = 100 is test/test_ABY3.cpp:155-180's size) and
    reveal_bits' xor (:287-288) against the oracle; the three parties' products reveal X & Y."""
    rng = np.random.default_rng(n)
    xs = [rng.integers(0, 2, n, dtype=np.int8) for _ in range(3)]
    ys = [rng.integers(0, 2, n, dtype=np.int8) for _ in range(3)]
    zs = []
    for j in range(3):
        nx = (j + 1) % 3
        g = gpu.aby3_vecbool_mul_local(xs[j], xs[nx], ys[j], ys[nx])
        assert np.array_equal(g, orc.aby3_vecbool_mul_local(xs[j], xs[nx], ys[j], ys[nx]))
        zs.append(g)
    assert np.array_equal(zs[0] ^ zs[1] ^ zs[2], (xs[0] ^ xs[1] ^ xs[2]) & (ys[0] ^ ys[1] ^ ys[2]))
    # inputs outside {0, 1}: the reference's int8 & and ^ apply to whole bytes
    wild = [rng.integers(-128, 128, n, dtype=np.int8) for _ in range(4)]
    assert np.array_equal(gpu.aby3_vecbool_mul_local(*wild), orc.aby3_vecbool_mul_local(*wild))
    a, b, c = rnd(rng, n), rnd(rng, n), rnd(rng, n)
    got = gpu.aby3_reveal3_bits(a, b, c)
    assert np.array_equal(got, orc.aby3_reveal3_bits(a, b, c)) and np.array_equal(got, a ^ b ^ c)


@pytest.mark.parametrize("n", [1, 15, 16, 17, 64, 1000, 70001])
def test_bit_inject_senders_with_unaligned_inputs(gpu, orc, n):
    """The OT senders' segments (parties 0 and 1, before the round) on operands that start 8 bytes (words) and 3 bytes
    (booleans) into their allocations, and on booleans that are not 0 / 1: the messages equal the oracle's."""
    ctx = gpu.ctx
    rng = np.random.default_rng(n)
    A0, A1 = rnd(rng, n), rnd(rng, n)
    for wild in (False, True):
        B0, B1 = (rng.integers(-128, 128, n, dtype=np.int8) if wild else rng.integers(0, 2, n, dtype=np.int8) for _ in range(2))
        dA0, dA1 = (ctx.to_device(np.concatenate([np.zeros(1, np.int64), a])).view(1, n) for a in (A0, A1))
        dB0, dB1 = (ctx.to_device(np.concatenate([np.zeros(3, np.int8), b])).view(3, n) for b in (B0, B1))
        o1, o2 = ctx.empty(np.int64, 2 * n + 1).view(1, 2 * n), ctx.empty(np.int64, 3 * n)
        ctx.call("aby3_bit_inject_p0_pre", dA0, dB0, dB1, n, 0, n, KEY01, 7, KEY20, 130, o1, o2)
        ctx.sync()
        w1, w2 = orc.aby3_bit_inject_p0_pre(A0, B0, B1, KEY01, 7, KEY20, 130)
        assert np.array_equal(o1.to_host(), w1) and np.array_equal(o2.to_host(), w2)
        ctx.call("aby3_bit_inject_p1_pre", dA0, dA1, dB0, dB1, n, 0, n, KEY01, 7, KEY12, 64, o1, o2)
        ctx.sync()
        w1, w2 = orc.aby3_bit_inject_p1_pre(A0, A1, B0, B1, KEY01, 7, KEY12, 64)
        assert np.array_equal(o1.to_host(), w1) and np.array_equal(o2.to_host(), w2)


@pytest.mark.parametrize("n", [1, 63, 4100])
def test_bit_inject_post_sub_in_place(gpu, orc, n):
    """mpcx_aby3_bit_inject_p*_post_sub with the result written over the minuend (R = S, as the header allows): equals
    the oracle's post segment followed by ring_sub (element ranges of the fused call: tests/test_gpu_ranges.py)."""
    ctx = gpu.ctx
    rng = np.random.default_rng(n)
    nt, i0 = n, 0
    B0, B1 = (rng.integers(0, 2, n, dtype=np.int8) for _ in range(2))
    S0, S1 = rnd(rng, n), rnd(rng, n)
    fa, fb, fc = rnd(rng, 3 * n), rnd(rng, 3 * n), rnd(rng, 3 * n)
    cases = [("aby3_bit_inject_p0_post_sub", (B0,), (KEY01, 5, KEY20, 9), fa[:2 * n], fb[:n],
              lambda: orc.aby3_bit_inject_p0_post(B0, KEY01, 5, KEY20, 9, fa[:2 * n], fb[:n])),
             ("aby3_bit_inject_p1_post_sub", (B1,), (KEY01, 5, KEY12, 9), fa[:2 * n], fb[:n],
              lambda: orc.aby3_bit_inject_p1_post(B1, KEY01, 5, KEY12, 9, fa[:2 * n], fb[:n])),
             ("aby3_bit_inject_p2_post_sub", (B0, B1), (KEY20, 5, KEY12, 9), fa, fc,
              lambda: orc.aby3_bit_inject_p2_post(B0, B1, KEY20, 5, KEY12, 9, fa, fc))]
    for name, bits, keys, m1, m2, want in cases:
        C = want()
        d0, d1 = ctx.to_device(S0), ctx.to_device(S1)
        ctx.call(name, *[ctx.to_device(b) for b in bits], nt, i0, n, *keys, ctx.to_device(m1), ctx.to_device(m2), d0, d1, d0, d1)
        ctx.sync()
        assert np.array_equal(d0.to_host(), orc.ring_sub(S0, C[0])), name
        assert np.array_equal(d1.to_host(), orc.ring_sub(S1, C[1])), name


def test_empty_inputs_are_noops(gpu):
    e = np.zeros(0, dtype=np.int64)
    assert gpu.ring_add(e, e).size == 0
    assert gpu.aby3_share_local(None, 0, KEY01, 0, KEY20, 0).size == 0
    assert gpu.aby3_reveal3(e, e, e).size == 0


def test_bad_shapes_are_rejected(gpu):
    from mpc_amd import MpcxError
    one = np.zeros(1, dtype=np.int64)
    with pytest.raises(MpcxError):
        gpu.ring_gemm(one, one, 0, 1, 1)
    with pytest.raises(MpcxError):
        gpu._go("aby3_msb_round_pre", np.zeros(4, np.uint64), 1, 3, ("out", np.int8, 8))
    b = np.zeros(4, dtype=np.int8)
    with pytest.raises(MpcxError, match="outside a vector"):  # elements [8, 12) of a 10-element vector
        gpu.aby3_bit_inject_p2_pre(b, b, KEY20, 0, KEY12, 0, rng=(10, 8))
    # the additions: prepared operands, bit-string segments, narrowed copies
    import ctypes
    ctx = gpu.ctx
    d = ctx.to_device(one)
    P, U64, op = lambda x: ctypes.c_void_p(x.ptr), ctypes.c_uint64, ctypes.c_void_p()
    with pytest.raises(MpcxError, match="empty shape"):
        ctx._call("mpcx_aby3_operand_prepare", P(d), P(d), U64(0), U64(1), ctypes.c_int(0), ctypes.byref(op))
    with pytest.raises(MpcxError, match="layout"):
        ctx._call("mpcx_aby3_operand_prepare", P(d), P(d), U64(1), U64(1), ctypes.c_int(2), ctypes.byref(op))
    with pytest.raises(MpcxError, match="NULL"):
        ctx._call("mpcx_aby3_matmul_local_prepared", P(d), P(d), ctypes.c_int(0), None, U64(1), U64(0), U64(KEY01), U64(0),
                  U64(KEY20), U64(0), P(d))
    ctx._call("mpcx_aby3_operand_free", None)  # nothing to free: accepted
    with pytest.raises(MpcxError, match="temp must be"):
        ctx._call("mpcx_aby3_msb_round_pre_bits", P(ctx.empty(np.uint64, 8)), U64(1), U64(3), P(ctx.empty(np.uint32, 2)))
    with pytest.raises(MpcxError, match="NULL"):
        ctx._call("mpcx_d2h_bits", None, P(d), U64(5))
    with pytest.raises(MpcxError, match="NULL"):
        ctx._call("mpcx_h2d_u16", P(d), None, U64(5))
    ctx._call("mpcx_h2d_bits", None, None, U64(0))  # empty payloads are no-ops
    ctx._call("mpcx_d2h_u16", None, None, U64(0))


@pytest.mark.parametrize("n", [1, 3, 13, 777, 5001])
def test_aby3_msb_messages_at_8_byte_alignment(gpu, orc, n, monkeypatch):
    """msb round messages that are 8- but not 16-byte aligned take the kernels' narrower load / store path."""
    from backends import GpuBackend
    monkeypatch.setattr(GpuBackend, "msb_msg_skew", 8)
    X = prg_pops(25, 0, n)
    res = []
    for be in (gpu, orc):
        w = pr.World(be)
        Gs, Ps = pr.aby3_add_prepare(w, pr.aby3_share(w, X, n))
        res.append((w, pr.aby3_msb(w, Gs, Ps)))
    same_logs(res[0][0], res[1][0])
    for j in range(3):
        for i in range(2):
            assert np.array_equal(res[0][1][j][i], res[1][1][j][i])


@pytest.mark.parametrize("nbits", [1, 7, 8, 63, 64, 65, 4099, (1 << 20) + 13, (36 << 20) + 77])
def test_boolean_payloads_cross_the_link_as_bits(gpu, nbits):
    """mpcx_h2d_bits / mpcx_d2h_bits: bytes of 0 / 1 on the host <-> a bit string on the device (several staging chunks
    for the largest), only bit 0 of a host byte counts, the bits of the last word beyond the payload arrive as zero."""
    import ctypes
    ctx = gpu.ctx
    rng = np.random.default_rng(nbits)
    src = rng.integers(0, 256, nbits, dtype=np.uint8)  # wild bytes: bit 0 is the boolean
    words = (nbits + 31) // 32
    dev = ctx.to_device(np.full(words, 0xFFFFFFFF, dtype=np.uint32))
    ctx._call("mpcx_h2d_bits", ctypes.c_void_p(dev.ptr), ctypes.c_void_p(src.ctypes.data), ctypes.c_uint64(nbits))
    got_words = dev.to_host()
    want_bits = np.zeros(words * 32, dtype=np.uint8)
    want_bits[:nbits] = src & 1
    assert np.array_equal(got_words, np.packbits(want_bits, bitorder="little").view(np.uint32))
    back = np.full(nbits + 3, 7, dtype=np.uint8)
    ctx._call("mpcx_d2h_bits", ctypes.c_void_p(back.ctypes.data), ctypes.c_void_p(dev.ptr), ctypes.c_uint64(nbits))
    assert np.array_equal(back[:nbits], src & 1) and np.all(back[nbits:] == 7)


@pytest.mark.parametrize("count,wide_at", [(1000, None), ((2 << 20) + 3, None), ((9 << 20) + 5, None), ((9 << 20) + 5, 0),
                                           ((9 << 20) + 5, (8 << 20) + 77), ((2 << 20) + 3, (2 << 20) + 2)])
@pytest.mark.parametrize("skew", [0, 1])
def test_16_bit_payloads_that_fit_a_byte(gpu, count, wide_at, skew):
    """mpcx_h2d_u16 / mpcx_d2h_u16 return exactly what mpcx_h2d / mpcx_d2h of the same array return: arrays of Z_67-like
    values go narrowed (several staging chunks for the largest), an array with one value above 255 anywhere goes as it
    is; host memory at an odd address, device arrays that start one element into a buffer."""
    import ctypes
    ctx = gpu.ctx
    rng = np.random.default_rng(count + skew)
    vals = rng.integers(0, 67, count, dtype=np.uint16)
    if wide_at is not None:
        vals[wide_at] = 256 + 66
    raw = np.zeros(2 * count + 8, dtype=np.uint8)
    raw[skew:skew + 2 * count] = vals.view(np.uint8)  # the host array at an odd address when skew = 1
    dev = ctx.to_device(np.full(count + 8, 0xABCD, dtype=np.uint16))
    dptr = dev.ptr + 2 * skew  # ... and the device array one element into its buffer
    ctx._call("mpcx_h2d_u16", ctypes.c_void_p(dptr), ctypes.c_void_p(raw.ctypes.data + skew), ctypes.c_uint64(count))
    on_device = dev.to_host()
    assert np.array_equal(on_device[skew:skew + count], vals)
    assert np.all(on_device[:skew] == 0xABCD) and np.all(on_device[skew + count:] == 0xABCD)
    back = np.full(2 * count + 8, 0x5A, dtype=np.uint8)
    ctx._call("mpcx_d2h_u16", ctypes.c_void_p(back.ctypes.data + skew), ctypes.c_void_p(dptr), ctypes.c_uint64(count))
    assert np.array_equal(back[skew:skew + 2 * count], vals.view(np.uint8))
    assert np.all(back[:skew] == 0x5A) and np.all(back[skew + 2 * count:] == 0x5A)


@pytest.mark.parametrize("n", [1, 5, 63, 64, 65, 1000, 4097, (1 << 18) + 4099])
def test_aby3_msb_segments_on_bit_strings(gpu, n):
    """The msb segments with messages kept as bit strings on the device (round_pre_bits / round_mid_bits / last_bits,
    bytes made and read on the host) against the byte forms: every outgoing message, the tree state they leave behind
    and both result vectors are identical, whatever bytes come in."""
    import ctypes
    ctx = gpu.ctx
    rng = np.random.default_rng(n)
    G0, G1, P0, P1 = (ctx.to_device(rnd(rng, n)) for _ in range(4))
    incoming = {t: rng.integers(0, 2, 2 * t * n, dtype=np.int8) for t in (32, 16, 8, 4, 2, 1)}
    U64, P = ctypes.c_uint64, lambda d: ctypes.c_void_p(d.ptr)

    def run(bits):
        state = ctx.empty(np.uint64, 6 * n)
        ctx.call("aby3_msb_begin", G0, G1, P0, P1, n, state)
        sent = []
        words = lambda nb: (nb + 31) // 32
        out_b, in_b = ctx.empty(np.int8, 64 * n), ctx.empty(np.int8, 64 * n)
        out_w, in_w = ctx.empty(np.uint32, words(64 * n)), ctx.empty(np.uint32, words(64 * n))
        Z0, Z1 = ctx.empty(np.int8, n), ctx.empty(np.int8, n)

        def fetch(nb):
            if not bits:
                return out_b.to_host()[:nb].copy()
            host = np.zeros(nb, dtype=np.int8)
            ctx._call("mpcx_d2h_bits", ctypes.c_void_p(host.ctypes.data), P(out_w), U64(nb))
            return host

        def feed(msg):
            if not bits:
                ctx._call("mpcx_h2d", P(in_b), ctypes.c_void_p(msg.ctypes.data), U64(msg.size))
            else:
                ctx._call("mpcx_h2d_bits", P(in_w), ctypes.c_void_p(msg.ctypes.data), U64(msg.size))

        if bits:
            ctx._call("mpcx_aby3_msb_round_pre_bits", P(state), U64(n), U64(32), P(out_w))
        else:
            ctx.call("aby3_msb_round_pre", state, n, 32, out_b)
        sent.append(fetch(64 * n))
        for t in (32, 16, 8, 4, 2):
            feed(incoming[t])
            if bits:
                ctx._call("mpcx_aby3_msb_round_mid_bits", P(state), U64(n), U64(t), P(in_w), P(out_w))
            else:
                ctx.call("aby3_msb_round_mid", state, n, t, in_b, out_b)
            sent.append(fetch(t * n))
        feed(incoming[1])
        if bits:
            ctx._call("mpcx_aby3_msb_last_bits", P(state), U64(n), P(in_w), P(Z0), P(Z1))
        else:
            ctx.call("aby3_msb_last", state, n, in_b, Z0, Z1)
        return sent, Z0.to_host(), Z1.to_host()

    want, got = run(False), run(True)
    for k, (a, b) in enumerate(zip(want[0], got[0])):
        assert np.array_equal(a, b), f"message {k} differs"
    assert np.array_equal(want[1], got[1]) and np.array_equal(want[2], got[2])


def test_profile_collects_per_family_and_select_restricts_it(gpu):
    """mpcx_profile / mpcx_profile_select / mpcx_profile_read: launches are counted per kernel family, a selected
    family is the only one that records, and the kernel-name query names the GEMM kernel a shape runs in."""
    import ctypes
    from mpc_amd import MpcxError, load_library
    ctx = gpu.ctx
    rng = np.random.default_rng(3)
    r, l, c = 130, 70, 66
    A, B = ctx.to_device(rnd(rng, r * l)), ctx.to_device(rnd(rng, l * c))
    Z = ctx.to_device(np.zeros(r * c, np.int64))

    def read(name):
        ms, cnt = ctypes.c_double(), ctypes.c_uint64()
        ctx._call("mpcx_profile_read", name.encode(), ctypes.byref(ms), ctypes.byref(cnt))
        return ms.value, cnt.value

    ctx.call("profile", 1)
    for _ in range(2):
        ctx.call("ring_gemm_u64", A, B, Z, r, l, c, 0, 0, 0, 0)
    ms, cnt = read("limb_gemm")
    assert cnt == 2 and ms > 0 and read("split_limbs")[1] >= 2
    ctx._call("mpcx_profile_select", b"limb_gemm")
    ctx.call("profile", 1)  # clears
    ctx.call("ring_gemm_u64", A, B, Z, r, l, c, 0, 0, 0, 0)
    assert read("limb_gemm")[1] == 1 and read("split_limbs")[1] == 0
    with pytest.raises(MpcxError, match="unknown kernel name"):
        ctx._call("mpcx_profile_select", b"no_such_kernel")
    ctx._call("mpcx_profile_select", None)
    ctx.call("profile", 0)
    name = load_library().mpcx_ring_gemm_kernel_name
    name.restype = ctypes.c_char_p
    U = ctypes.c_uint64
    assert name(U(r), U(l), U(c)) == b"limb_gemm_kernel"
    assert name(U(4096), U(8192), U(4096)) in (b"rns_gemm_wide_kernel", b"rns_gemm_kernel")
