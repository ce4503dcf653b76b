"""The benchmark configurations at their full sizes, checked through properties that do not need the CPU oracle
(which would take hours there): reconstruction identities, exact sampled dot products, a Freivalds check of the
whole product, and the protocols' defining relations on every element.  Everything stays resident on the GPU."""
import os
import sys

import numpy as np
import pytest

import protocols as pr

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

KEY01, KEY20, KEYS = 0xAB01, 0xCA20, 0xF0


@pytest.fixture(scope="module")
def env():
    import torch
    from backends import ResidentBackend
    from mpc_amd import Context
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ctx = Context(0)
    ctx.call("set_stream", torch.cuda.current_stream().cuda_stream)
    yield ctx, torch, dev, ResidentBackend(ctx, torch, dev)
    torch.cuda.synchronize()
    ctx.call("set_rand_cache", 0)
    torch.cuda.empty_cache()


def test_aby3_share_matmul_4096(env):
    """ABY3 4096 x 4096 share matmul + reconstruct, inputs from PRgenerator(20,0): the three parties reveal the same
    matrix, 24 sampled entries equal exact big-integer dot products, and Z v = X (Y v) over Z_2^64."""
    ctx, torch, dev, _ = env
    from bench import Aby3MatmulJob
    job = Aby3MatmulJob(ctx, torch, dev, 4096, 0, 1)
    job.step()
    torch.cuda.synchronize()
    job.verify()
    # the three replicated shares differ from one another and from the plain product (they are masked)
    assert not torch.equal(job.Z0[0], job.Z0[1]) and not torch.equal(job.Z0[0], job.Zplain[0])


@pytest.mark.parametrize("n", [1, 77, 5000, 70001])
def test_aby3_relu_chain_with_bit_string_messages(env, orc, n):
    """The resident ReLU chain with the carry tree's messages handed from party to party as bit strings (the *_bits
    segments) gives the oracle's shares, word for word: the representation of a message between two device-resident
    parties changes nothing a party computes."""
    import protocols as pr
    ctx, torch, dev, be = env
    rng = np.random.default_rng(n)
    x = rng.integers(-2 ** 63, 2 ** 63, n, dtype=np.int64)

    def chain(backend, xin):
        w = pr.World(backend, record=False)
        Xs = pr.aby3_share(w, xin, n)
        Gs, Ps = pr.aby3_add_prepare(w, Xs)
        Zb = pr.aby3_msb(w, Gs, Ps)
        return pr.aby3_bit_inject(w, Xs, Zb, minuend=Xs)

    want = chain(orc, x)
    be.msb_bits = True
    try:
        got = chain(be, ctx.to_device(x))
    finally:
        be.msb_bits = False
    for j in range(3):
        for k in range(2):
            assert np.array_equal(be.down(got[j][k]), want[j][k]), f"share {k} of party {j}"


def uless(torch, a, b):
    """a < b for int64 tensors read as unsigned."""
    m = torch.iinfo(torch.int64).min
    return (a ^ m) < (b ^ m)


def test_private_compare_2_24(env):
    """privateCompare on 2^24 elements drawn as test/test_securenn.cpp:160-175 draws them from PRgenerator(22,0) (x, r,
    beta, then the 64 Z_67 shares of party 0; r = -1 on every fifth element): beta xor beta' = (x > r) for every
    element, with the byte cache on and with it off."""
    ctx, torch, dev, _ = env
    from bench import private_compare_inputs
    n = 1 << 24
    ctx.call("prg_ckpt_build", 22, 0, 67 * n, 0)
    s0, s1, beta, r = private_compare_inputs(ctx, torch, dev, 0, n)
    x = torch.empty(n, dtype=torch.int64, device=dev)
    step = 1 << 18
    for a in range(0, n, step):  # x = the first of an element's 67 pops
        w = torch.empty(67 * step, dtype=torch.int64, device=dev)
        ctx.call("prg_fill_pop", 22, 0, 67 * a, 67 * step, w)
        x[a:a + step] = w.view(step, 67)[:, 0]
    del w
    assert int((r == -1).sum()) >= n // 5 and int(s0.max()) < 67 and int(s1.max()) < 67
    ctx.call("prg_ckpt_build", KEY01, 0, 0, 128 * n)
    want = uless(torch, r, x).to(torch.int8)
    d0, d1 = torch.empty_like(s0), torch.empty_like(s0)
    b2 = torch.empty(n, dtype=torch.int8, device=dev)
    for cap in (4 << 30, 0):
        ctx.call("set_rand_cache", cap)
        ctx.call("snn_pc_party", 0, s0, beta, r, n, 0, n, KEY01, d0)
        ctx.call("snn_pc_party", 1, s1, beta, r, n, 0, n, KEY01, d1)
        ctx.call("snn_pc_helper", d0, d1, n, b2)
        torch.cuda.synchronize()
        assert int(d0.max()) < 67 and int(d1.max()) < 67 and int(d0.min()) >= 0
        assert torch.equal(beta ^ b2, want), f"byte cache {cap}"
    ctx.call("set_rand_cache", 4 << 30)


def test_snn_msb_2_24(env):
    """SecureNN msb (shareConvert + msb_L_1, all three roles) on 2^24 values a = pop / 2: the two shares add up to the
    sign bit of every element."""
    ctx, torch, dev, be = env
    n = 1 << 24
    ctx.call("prg_ckpt_build", KEYS, 0, 0, 67 * n + 64)
    a = torch.empty(n, dtype=torch.int64, device=dev)
    ctx.call("prg_fill_pop", 22, 0, 0, n, a)
    a = torch.div(a, 2, rounding_mode="trunc")  # test/test_securenn.cpp:292-293
    w = pr.World(be, record=False)
    As = pr.snn_share(w, ctx.wrap(a, np.int64), n)
    B = pr.snn_msb(w, As[:2])
    got = torch.from_numpy(be.down(B[0])).to(dev) + torch.from_numpy(be.down(B[1])).to(dev)
    assert torch.equal(got, (a < 0).to(torch.int64))


def test_aby3_relu_chain_2_24(env):
    """ABY3 add_prepare -> msb -> bit_inject on 2^24 pops: x - x * msb(x) reveals max(x, 0), and the boolean shares of
    msb xor to the sign bit."""
    ctx, torch, dev, be = env
    n = 1 << 24
    x = torch.empty(n, dtype=torch.int64, device=dev)
    ctx.call("prg_fill_pop", 21, 0, 0, n, x)
    w = pr.World(be, record=False)
    Xs = pr.aby3_share(w, ctx.wrap(x, np.int64), n)
    Gs, Ps = pr.aby3_add_prepare(w, Xs)
    Zb = pr.aby3_msb(w, Gs, Ps)
    C = pr.aby3_bit_inject(w, Xs, Zb)
    up = lambda d: torch.from_numpy(be.down(d)).to(dev)
    sign = up(Zb[0][0]) ^ up(Zb[1][0]) ^ up(Zb[2][0])
    assert torch.equal(sign, (x < 0).to(torch.int8))
    relu = sum(up(be.ring_sub(Xs[j][0], C[j][0])) for j in range(3))
    assert torch.equal(relu, torch.clamp(x, min=0))
    for j in range(3):  # replicated sharing: party j's second share is party j+1's first
        assert torch.equal(up(C[j][1]), up(C[(j + 1) % 3][0]))


def test_aby3_relu_chain_2_28_in_batches(env):
    """The 2^28-element ReLU chain as sixteen element ranges of 2^24 (n_total, i0, n), every one of them: each range
    reveals max(x, 0) although its pop-stream offsets are those of the whole vector (up to 7 * 2^28 words into each
    stream), and the replicated shares of neighbouring parties agree."""
    ctx, torch, dev, be = env
    nt, n = 1 << 28, 1 << 24
    x = torch.empty(n, dtype=torch.int64, device=dev)
    T = be.tensor
    for b in range(16):
        i0 = b * n
        ctx.call("prg_fill_pop", 21, 0, i0, n, x)
        w = pr.World(be, record=False, n_total=nt, i0=i0)
        Xs = pr.aby3_share(w, ctx.wrap(x, np.int64), n)
        Gs, Ps = pr.aby3_add_prepare(w, Xs)
        Zb = pr.aby3_msb(w, Gs, Ps)
        C = pr.aby3_bit_inject(w, Xs, Zb)
        relu = sum(T(be.ring_sub(Xs[j][0], C[j][0])) for j in range(3))
        torch.cuda.synchronize()
        assert torch.equal(relu, torch.clamp(x, min=0)), f"range {b}"
        assert torch.equal(T(Zb[0][0]) ^ T(Zb[1][0]) ^ T(Zb[2][0]), (x < 0).to(torch.int8)), f"range {b}: sign bits"
        for j in range(3):
            assert torch.equal(T(C[j][1]), T(C[(j + 1) % 3][0])), f"range {b}: replicated shares disagree"
        del Xs, Gs, Ps, Zb, C, relu


def test_aby3_matmul_config5_row_block(env, monkeypatch):
    """One GPU's row block of the 16384 x 16384 configuration (rows [12288, 14336) of X, l = c = 16384: four
    accumulation chunks per product, mask words from offset row0 * c of both pop streams), computed by the digit-pair
    and by the residue kernels.  The two results are equal word for word; after the mask pop(next) - pop(prev)
    (ABY3Engine.cpp:190-197) is taken off, the block passes a Freivalds check, 24 entries equal exact big-integer dot
    products, and one whole row equals X0 (Y0 + Y1) + X1 Y0 (:189) computed with wrapping 64-bit arithmetic."""
    ctx, torch, dev, _ = env
    rows, l, c, row0, pm = 2048, 16384, 16384, 6 * 2048, 3 * 16384 * 16384 + 5
    e = lambda k: torch.empty(k, dtype=torch.int64, device=dev)
    X0, X1, Y0, Y1 = e(rows * l), e(rows * l), e(l * c), e(l * c)
    for k, t in enumerate((X0, X1, Y0, Y1)):  # uniform 64-bit words
        ctx.call("prg_fill_pop", 31 + k, 0, 0, t.numel(), t)
    Z = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("MPCX_GEMM_RNS", mode)
        Z[mode] = e(rows * c)
        ctx.call("aby3_matmul_local_rows", X0, X1, Y0, Y1, rows, l, c, row0, KEY01, pm, KEY20, pm, Z[mode])
    monkeypatch.delenv("MPCX_GEMM_RNS")
    torch.cuda.synchronize()
    assert torch.equal(Z["0"], Z["2"]), "digit-pair and residue kernels disagree"
    ma, mb = e(rows * c), e(rows * c)
    ctx.call("prg_fill_pop", KEY01, 0, pm + row0 * c, rows * c, ma)
    ctx.call("prg_fill_pop", KEY20, 0, pm + row0 * c, rows * c, mb)
    torch.cuda.synchronize()
    P = (Z["0"] - (ma - mb)).view(rows, c)  # the product without its mask
    del ma, mb, Z
    X0m, X1m, Y0m, Ysum = X0.view(rows, l), X1.view(rows, l), Y0.view(l, c), (Y0 + Y1).view(l, c)
    matvec = lambda M, v: (M * v.unsqueeze(0)).sum(1)  # int64 arithmetic on the device wraps mod 2^64
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    for _ in range(3):
        v = torch.randint(-2 ** 63, 2 ** 63 - 1, (c,), dtype=torch.int64, device=dev, generator=g) | 1
        assert torch.equal(matvec(P, v), matvec(X0m, matvec(Ysum, v)) + matvec(X1m, matvec(Y0m, v))), "P v != X0 ((Y0+Y1) v) + X1 (Y0 v)"
    i = 1357  # one whole row
    want_row = (X0m[i].unsqueeze(1) * Ysum).sum(0) + (X1m[i].unsqueeze(1) * Y0m).sum(0)
    assert torch.equal(P[i], want_row), "row of the product differs"
    rng = np.random.default_rng(9)
    for _ in range(24):  # exact integers on the host
        i, k = int(rng.integers(rows)), int(rng.integers(c))
        x0, x1 = X0m[i].cpu().numpy().view(np.uint64), X1m[i].cpu().numpy().view(np.uint64)
        ys, y0 = Ysum[:, k].cpu().numpy().view(np.uint64), Y0m[:, k].cpu().numpy().view(np.uint64)
        want = (sum(int(a) * int(b) for a, b in zip(x0, ys)) + sum(int(a) * int(b) for a, b in zip(x1, y0))) % (1 << 64)
        assert int(P[i, k].item()) % (1 << 64) == want, f"P[{i},{k}] wrong"


@pytest.mark.parametrize("r,l,c", [(1024, 2048, 1024), (2560, 2048, 2560)])
def test_ring_gemm_auto_selected_residue_form_vs_oracle(env, orc, r, l, c):
    """Products for which the library picks the residue kernels by itself (the smallest cube-like shape where its
    estimate favours them, and the 2560-wide one), word for word against the CPU oracle (row blocks of the oracle's
    product on eight host threads)."""
    from concurrent.futures import ThreadPoolExecutor
    import mpc_amd
    ctx, torch, dev, _ = env
    import ctypes
    nprod = lambda *dims: mpc_amd.load_library().mpcx_ring_gemm_products(*[ctypes.c_uint64(d) for d in dims])
    assert nprod(r, l, c) == 18
    # short contractions, thin outputs and small products stay with the digit pairs
    assert nprod(2560, 512, 2560) == 36 and nprod(128, 8192, 8192) == 36 and nprod(300, 300, 300) == 36
    rng = np.random.default_rng(r)
    A = rng.integers(-2 ** 63, 2 ** 63, (r, l), dtype=np.int64)
    B = rng.integers(-2 ** 63, 2 ** 63, (l, c), dtype=np.int64)
    dA, dB = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    dZ = torch.empty(r * c, dtype=torch.int64, device=dev)
    ctx.call("ring_gemm_u64", dA, dB, dZ, r, l, c, 0, 0, 0, 0)
    torch.cuda.synchronize()
    blk = r // 8
    with ThreadPoolExecutor(8) as ex:
        parts = list(ex.map(lambda b: orc.ring_gemm(A[b * blk:(b + 1) * blk].ravel(), B.ravel(), blk, l, c), range(8)))
    assert np.array_equal(dZ.cpu().numpy(), np.concatenate(parts))


@pytest.mark.parametrize("lay,mode", [(0, "0"), (1, "0"), (0, "2"), (1, "2")])
def test_ring_gemm_ragged_large(env, lay, mode, monkeypatch):
    """A 3001 x 5003 . 5003 x 2005 ring product (no dimension a multiple of the tile sizes; one accumulation chunk), in
    both layouts and accumulating onto a previous result: Z v = C v + A (B v) over Z_2^64."""
    monkeypatch.setenv("MPCX_GEMM_RNS", mode)  # "0": digit-pair kernel, "2": residue kernel
    ctx, torch, dev, _ = env
    r, l, c = 3001, 5003, 2005
    rng = np.random.default_rng(77 + lay)
    A = rng.integers(-2 ** 63, 2 ** 63, (r, l), dtype=np.int64)
    B = rng.integers(-2 ** 63, 2 ** 63, (l, c), dtype=np.int64)
    C = rng.integers(-2 ** 63, 2 ** 63, (r, c), dtype=np.int64)
    flat = (lambda M: np.ascontiguousarray(M.T).ravel()) if lay else (lambda M: M.ravel())
    dA, dB, dZ = (torch.from_numpy(flat(M).copy()).to(dev) for M in (A, B, C))
    ctx.call("ring_gemm_u64", dA, dB, dZ, r, l, c, lay, lay, lay, 1)
    torch.cuda.synchronize()
    Z = dZ.cpu().numpy()
    Z = Z.reshape(c, r).T if lay else Z.reshape(r, c)
    with np.errstate(over="ignore"):
        Au, Bu, Cu, Zu = (M.view(np.uint64) if M.flags.c_contiguous else np.ascontiguousarray(M).view(np.uint64)
                          for M in (A, B, C, Z))
        for _ in range(4):
            v = rng.integers(0, 1 << 64, c, dtype=np.uint64) | np.uint64(1)
            assert np.array_equal(Zu @ v, Cu @ v + Au @ (Bu @ v))
