"""Randomised parity campaign (run by hand on a GPU box: `python tests/fuzz_parity.py [iterations] [seed]`; not collected
by pytest).  Every iteration draws one case -- a ring product of random shape / layouts / formulation / block tile, an ABY3 product with a
prepared second operand, an ABY3 ReLU chain, a SecureNN msb or a MyEngine ReLU chain of random length, whole or cut into random element ranges,
with the fused or the separate segment calls and the generator-byte cache on or off -- runs it on the HIP path and on
the CPU oracle and requires identical messages and shares."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import my_protocols as mp  # noqa: E402
import protocols as pr  # noqa: E402
from backends import GpuBackend, OracleBackend  # noqa: E402
from my_backends import GpuMy, OracleMy  # noqa: E402
from oracle_lib import prg_pops  # noqa: E402
from range_helpers import aby3_relu, run_in_ranges, same_logs, snn_msb  # noqa: E402


def rnd(rng, n):
    return rng.integers(-2 ** 63, 2 ** 63, int(n), dtype=np.int64)


def random_ranges(rng, n):
    """Cuts [0, n) into 1..4 contiguous ranges (some possibly of one element)."""
    k = int(rng.integers(1, 5))
    cuts = sorted(set(int(c) for c in rng.integers(1, max(2, n), k - 1))) if n > 1 else []
    edges = [0] + [c for c in cuts if 0 < c < n] + [n]
    return tuple((a, b - a) for a, b in zip(edges, edges[1:]))


def case_gemm(rng, gpu, orc):
    r, l, c = (int(v) for v in rng.integers(1, 701, 3))
    if rng.random() < 0.2:
        l = int(rng.integers(700, 3000))
    la, lb, lc, acc = (int(v) for v in rng.integers(0, 2, 4))
    os.environ["MPCX_GEMM_RNS"] = str(rng.choice(["0", "2", "1"]))
    os.environ["MPCX_RNS_TILE"] = str(rng.choice(["0", "1"]))
    A, B = rnd(rng, r * l), rnd(rng, l * c)
    if rng.random() < 0.3:  # extreme digits
        A[:: max(1, A.size // 50)] = np.int64(-0x7F7F7F7F7F7F7F80)
        B[:: max(1, B.size // 50)] = np.int64(0x7F7F7F7F7F7F7F7F)
    C = rnd(rng, r * c) if acc else None
    got = gpu.ring_gemm(A, B, r, l, c, la, lb, lc, C=C)
    want = orc.ring_gemm(A, B, r, l, c, la, lb, lc, C=C)
    assert np.array_equal(got, want), ("gemm", r, l, c, la, lb, lc, acc, os.environ["MPCX_GEMM_RNS"], os.environ["MPCX_RNS_TILE"])
    return f"gemm {r}x{l}x{c} lay {la}{lb}{lc} acc {acc} rns {os.environ['MPCX_GEMM_RNS']} tile {os.environ['MPCX_RNS_TILE']}"


def case_prepared_matmul(rng, gpu, orc):
    """ABY3 matrix_mul segment with the second operand prepared once and used by two or three products of different row
    counts and row offsets, the block tile possibly switched in between (the planes then no longer fit)."""
    import ctypes
    from oracle_lib import prg_pops
    ctx = gpu.ctx
    l, c = int(rng.integers(1, 900)), int(rng.integers(1, 700))
    if rng.random() < 0.25:
        l = int(rng.integers(900, 6000))
    lay_x, lay_y = (int(v) for v in rng.integers(0, 2, 2))
    os.environ["MPCX_GEMM_RNS"] = str(rng.choice(["0", "2", "2", "1"]))
    os.environ["MPCX_RNS_TILE"] = str(rng.choice(["0", "1"]))
    to_lay = lambda M, a, b, lay: np.ascontiguousarray(M.reshape(a, b).T).ravel() if lay else M
    Y0, Y1 = rnd(rng, l * c), rnd(rng, l * c)
    dY0, dY1 = ctx.to_device(to_lay(Y0, l, c, lay_y)), ctx.to_device(to_lay(Y1, l, c, lay_y))
    op, U64 = ctypes.c_void_p(), ctypes.c_uint64
    ctx._call("mpcx_aby3_operand_prepare", ctypes.c_void_p(dY0.ptr), ctypes.c_void_p(dY1.ptr), U64(l), U64(c),
              ctypes.c_int(lay_y), ctypes.byref(op))
    try:
        for _ in range(int(rng.integers(2, 4))):
            if rng.random() < 0.3:
                os.environ["MPCX_RNS_TILE"] = str(rng.choice(["0", "1"]))
            rows, row0 = int(rng.integers(1, 500)), int(rng.integers(0, 50))
            pn, pp = int(rng.integers(0, 1000)), int(rng.integers(0, 1000))
            X0, X1 = rnd(rng, rows * l), rnd(rng, rows * l)
            dX0, dX1 = ctx.to_device(to_lay(X0, rows, l, lay_x)), ctx.to_device(to_lay(X1, rows, l, lay_x))
            Z = ctx.empty(np.int64, rows * c)
            ctx._call("mpcx_aby3_matmul_local_prepared", ctypes.c_void_p(dX0.ptr), ctypes.c_void_p(dX1.ptr),
                      ctypes.c_int(lay_x), op, U64(rows), U64(row0), U64(0xAB01), U64(pn), U64(0xCA20), U64(pp),
                      ctypes.c_void_p(Z.ptr))
            mask = prg_pops(0xAB01, pn + row0 * c, rows * c) - prg_pops(0xCA20, pp + row0 * c, rows * c)
            with np.errstate(over="ignore"):
                want = orc.ring_gemm(X0, Y0 + Y1, rows, l, c) + orc.ring_gemm(X1, Y0, rows, l, c) + mask
            assert np.array_equal(Z.to_host(), want), ("prepared", rows, l, c, lay_x, lay_y, row0, os.environ["MPCX_GEMM_RNS"])
    finally:
        ctx._call("mpcx_aby3_operand_free", op)
    return f"prepared matmul l {l} c {c} lay {lay_x}{lay_y} rns {os.environ['MPCX_GEMM_RNS']}"


def case_aby3_relu(rng, gpu, orc):
    n = int(rng.integers(1, 30000))
    pr.MSB_FUSED = bool(rng.integers(0, 2))
    x = rnd(rng, n)
    x[: min(n, 6)] = np.array([0, -1, 1, -2 ** 63, 2 ** 63 - 1, -2], dtype=np.int64)[: min(n, 6)]
    wo = pr.World(orc)
    want = aby3_relu(wo, x)
    ranges = random_ranges(rng, n)
    got, log = run_in_ranges(gpu, aby3_relu, x, ranges, halves=4)
    same_logs(log, wo.log)
    for j in range(3):
        assert np.array_equal(got[j], want[j]), ("aby3 relu", n, ranges, j)
    return f"aby3 relu n {n} ranges {ranges} fused {pr.MSB_FUSED}"


def case_snn_msb(rng, gpu, orc):
    n = int(rng.integers(1, 20000))
    pr.SNN_FUSED = bool(rng.integers(0, 2))
    cache = bool(rng.integers(0, 2))
    gpu.ctx.call("set_rand_cache", (4 << 30) if cache else 0)
    a = rnd(rng, n) >> 1
    a[: min(n, 4)] = np.array([0, -1, -(1 << 62), (1 << 62) - 1], dtype=np.int64)[: min(n, 4)]
    wo = pr.World(orc)
    want = snn_msb(wo, a)
    ranges = random_ranges(rng, n)
    got, log = run_in_ranges(gpu, snn_msb, a, ranges)
    same_logs(log, wo.log)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), ("snn msb", n, ranges)
    assert np.array_equal((got[0].view(np.uint64) + got[1].view(np.uint64)).view(np.int64), (a < 0).astype(np.int64))
    return f"snn msb n {n} ranges {ranges} fused {pr.SNN_FUSED} cache {cache}"


def case_pc(rng, gpu, orc):
    n = int(rng.integers(1, 5000))
    cache = bool(rng.integers(0, 2))
    gpu.ctx.call("set_rand_cache", (4 << 30) if cache else 0)
    hi = 67 if rng.random() < 0.5 else 65536  # the protocol's domain, or arbitrary 16-bit shares
    xb = rng.integers(0, hi, 64 * n, dtype=np.uint16)
    beta = rng.integers(0, 2, n, dtype=np.int8) if hi == 67 else rng.integers(-2, 3, n, dtype=np.int8)
    r = rnd(rng, n)
    r[:: int(rng.integers(2, 9))] = -1
    nt = n + int(rng.integers(0, 1000))
    i0 = int(rng.integers(0, nt - n + 1))
    from oracle_lib import KEY01
    for party in (0, 1):
        assert np.array_equal(gpu.snn_pc_party(party, xb, beta, r, nt, i0, KEY01),
                              orc.snn_pc_party(party, xb, beta, r, nt, i0, KEY01)), ("pc", n, nt, i0, hi, cache, party)
    return f"pc n {n} of {nt} at {i0} domain {hi} cache {cache}"


def case_my_relu(rng, gmy, omy):
    n = int(rng.integers(1, 20000))
    X = rnd(rng, n) // 2
    res = []
    for be in (gmy, omy):
        w = pr.World(be)
        Xs = mp.my_share(w, X, n)
        A = mp.my_msb(w, Xs)
        Z = mp.my_bit_inject(w, Xs, A)
        res.append((w, A, Z))
    same_logs(res[0][0].log, res[1][0].log)
    for j in range(2):
        assert np.array_equal(res[0][1][j], res[1][1][j]) and np.array_equal(res[0][2][j], res[1][2][j]), ("my relu", n, j)
    return f"my relu n {n}"


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    gpu, orc, gmy, omy = GpuBackend(), OracleBackend(), GpuMy(), OracleMy()
    cases = [(case_gemm, (gpu, orc), 0.3), (case_prepared_matmul, (gpu, orc), 0.1), (case_aby3_relu, (gpu, orc), 0.2),
             (case_snn_msb, (gpu, orc), 0.2), (case_pc, (gpu, orc), 0.12), (case_my_relu, (gmy, omy), 0.08)]
    p = np.array([c[2] for c in cases])
    t0 = time.time()
    for it in range(iters):
        fn, args, _ = cases[int(rng.choice(len(cases), p=p / p.sum()))]
        desc = fn(rng, *args)
        if it % 10 == 0 or time.time() - t0 > 30:
            print(f"[{it}] ok: {desc}", flush=True)
            t0 = time.time()
    for k in ("MPCX_GEMM_RNS", "MPCX_RNS_TILE"):
        os.environ.pop(k, None)
    print(f"fuzz_parity: {iters} cases passed (seed {seed})", flush=True)


if __name__ == "__main__":
    main()
