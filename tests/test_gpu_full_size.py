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

KEY01, KEYS = 0xAB01, 0xF0


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


def uless(torch, a, b):
    """a < b for int64 tensors read as unsigned."""
    m = torch.iinfo(torch.int64).min
    return (a ^ m) < (b ^ m)


def test_private_compare_2_24(env):
    """privateCompare on 2^24 elements (r = -1 on every fifth): beta xor beta' = (x > r) for every element."""
    ctx, torch, dev, _ = env
    n = 1 << 24
    g = torch.Generator(device=dev)
    g.manual_seed(22)
    x = torch.randint(-2 ** 63, 2 ** 63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
    r = torch.randint(-2 ** 63, 2 ** 63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
    r[2::5] = -1
    beta = torch.randint(0, 2, (n,), dtype=torch.int8, device=dev, generator=g)
    s0 = torch.randint(0, 67, (n, 64), dtype=torch.int16, device=dev, generator=g)
    s1 = torch.empty_like(s0)
    k = torch.arange(64, device=dev)
    step = 1 << 20
    for a in range(0, n, step):  # bit k of x, shared over Z_67 (test/test_securenn.cpp:167-175)
        bits = ((x[a:a + step].unsqueeze(1) >> k) & 1).to(torch.int16)
        s1[a:a + step] = (bits + 67 - s0[a:a + step]) % 67
    del bits
    ctx.call("prg_ckpt_build", KEY01, 0, 0, 128 * n)
    ctx.call("set_rand_cache", 4 << 30)
    d0, d1 = torch.empty_like(s0), torch.empty_like(s0)
    b2 = torch.empty(n, dtype=torch.int8, device=dev)
    ctx.call("snn_pc_party", 0, s0, beta, r, n, 0, n, KEY01, d0)
    ctx.call("snn_pc_party", 1, s1, beta, r, n, 0, n, KEY01, d1)
    ctx.call("snn_pc_helper", d0, d1, n, b2)
    torch.cuda.synchronize()
    assert int(d0.max()) < 67 and int(d1.max()) < 67 and int(d0.min()) >= 0
    want = uless(torch, r, x).to(torch.int8)
    assert torch.equal(beta ^ b2, want)


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
    """The 2^28-element ReLU chain as sixteen element ranges of 2^24 (n_total, i0, n): every range reveals max(x, 0)
    although its pop-stream offsets are those of the whole vector (up to 7 * 2^28 words into each stream)."""
    ctx, torch, dev, be = env
    nt, n = 1 << 28, 1 << 24
    x = torch.empty(n, dtype=torch.int64, device=dev)
    up = lambda d: torch.from_numpy(be.down(d)).to(dev)
    for b in (0, 7, 15):  # first, middle and last range (each is independent of the others)
        i0 = b * n
        ctx.call("prg_fill_pop", 21, 0, i0, n, x)
        w = pr.World(be, record=False, n_total=nt, i0=i0)
        Xs = pr.aby3_share(w, ctx.wrap(x, np.int64), n)
        Gs, Ps = pr.aby3_add_prepare(w, Xs)
        Zb = pr.aby3_msb(w, Gs, Ps)
        C = pr.aby3_bit_inject(w, Xs, Zb)
        relu = sum(up(be.ring_sub(Xs[j][0], C[j][0])) for j in range(3))
        assert torch.equal(relu, torch.clamp(x, min=0)), f"range {b}"
        for j in range(3):
            assert torch.equal(up(C[j][1]), up(C[(j + 1) % 3][0])), f"range {b}: replicated shares disagree"


@pytest.mark.parametrize("lay,mode", [(0, "0"), (1, "0"), (0, "2"), (1, "2")])
def test_ring_gemm_ragged_large(env, lay, mode, monkeypatch):
    """A 3001 x 5003 . 5003 x 2005 ring product (no dimension a multiple of the tile sizes; two K chunks when the parts
    are concatenated), in both layouts and accumulating onto a previous result: Z v = C v + A (B v) over Z_2^64."""
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
