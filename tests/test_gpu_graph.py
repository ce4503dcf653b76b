"""The library's calls under HIP graph capture.  After a warm-up pass a context allocates nothing and synchronises nothing
inside its calls (workspace, generator tables and the byte cache are in place), so a whole protocol -- all three roles,
messages handed over as device pointers -- can be captured from the stream once and replayed as one submission.  The
replay must give what the eager pass gives, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from backends import ResidentBackend
    from mpc_amd import Context
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ctx = Context(0)
    side = torch.cuda.Stream(dev)
    ctx.call("set_stream", side.cuda_stream)
    yield ctx, torch, dev, ResidentBackend(ctx, torch, dev), side
    torch.cuda.synchronize()
    ctx.call("set_rand_cache", 0)
    ctx.close()
    torch.cuda.empty_cache()


def capture(torch, side, fn):
    """fn() twice eagerly on `side` (the second result is kept), then once under capture; returns (eager, graph, captured)."""
    with torch.cuda.stream(side):
        fn()
        eager = fn()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        captured = fn()
    torch.cuda.synchronize()
    return eager, g, captured


def test_set_stream_to_the_same_stream_is_a_no_op_under_capture(env):
    """mpcx_set_stream synchronises the stream it leaves; asked for the stream the context already uses it must touch
    nothing -- a synchronisation would invalidate a capture in progress on that stream."""
    ctx, torch, dev, be, side = env
    x = torch.arange(1000, dtype=torch.int64, device=dev)
    y = torch.empty_like(x)
    with torch.cuda.stream(side):
        ctx.call("ring_shl1", x, y, 1000)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        ctx.call("set_stream", torch.cuda.current_stream().cuda_stream)
        ctx.call("ring_shl1", x, y, 1000)
    y.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, x << 1)


@pytest.mark.parametrize("n", [1000, 70001])
def test_aby3_relu_chain_replayed_from_a_graph(env, n):
    """share -> add_prepare -> msb -> bit_inject -> subtract for the three parties, captured and replayed twice."""
    ctx, torch, dev, be, side = env
    import protocols as pr
    rng = np.random.default_rng(n)
    x = rng.integers(-2 ** 63, 2 ** 63, n, dtype=np.int64)
    xD = ctx.to_device(x)

    def chain():
        w = pr.World(be, record=False)
        Xs = pr.aby3_share(w, xD, n)
        Gs, Ps = pr.aby3_add_prepare(w, Xs)
        Zb = pr.aby3_msb(w, Gs, Ps)
        R = pr.aby3_bit_inject(w, Xs, Zb, minuend=Xs)
        return [R[j][0] for j in range(3)]

    eager, g, captured = capture(torch, side, chain)
    want = [be.down(e).copy() for e in eager]
    assert np.array_equal(want[0] + want[1] + want[2], np.maximum(x, 0))
    for _ in range(2):
        for c in captured:
            be.tensor(c).zero_()
        g.replay()
        torch.cuda.synchronize()
        for j in range(3):
            assert np.array_equal(be.down(captured[j]), want[j]), f"share {j} of the replay differs from the eager pass"


def test_snn_matmul_msb_layer_replayed_from_a_graph(env):
    """One SecureNN layer of the MLP configuration (matrix_mul -> >> 16 -> msb -> vector_mul -> subtract on 128 x 784 by
    784 x 128, which reads generator buffers beyond the reference's BUFF_NUM), captured and replayed."""
    ctx, torch, dev, be, side = env
    import copy

    import protocols as pr
    r, l, c = 128, 784, 128
    rng = np.random.default_rng(4)
    X = rng.integers(-2 ** 15, 2 ** 15, r * l, dtype=np.int64)
    W = rng.integers(-2 ** 15, 2 ** 15, l * c, dtype=np.int64)
    ctx.call("set_rand_cache", 1 << 30)
    w = pr.World(be, record=False)
    with torch.cuda.stream(side):
        Xs = pr.snn_share(w, ctx.to_device(X), X.size)[:2]
        Ws = pr.snn_share(w, ctx.to_device(W), W.size)[:2]
    pops0 = copy.deepcopy(w.pops)

    def layer():
        w.pops = copy.deepcopy(pops0)
        Z = pr.snn_matmul(w, Xs, Ws, r, l, c)
        a = [be.ring_sar(Z[0], 16), be.ring_sar(Z[1], 16)]
        m = pr.snn_msb(w, a)
        am = pr.snn_vecmul(w, a, m)
        return [be.ring_sub(a[j], am[j]) for j in range(2)]

    eager, g, captured = capture(torch, side, layer)
    want = [be.down(e).copy() for e in eager]
    plain = np.maximum((X.reshape(r, l) @ W.reshape(l, c)) >> 16, 0)
    assert np.abs((want[0] + want[1]).reshape(r, c) - plain).max() <= 2  # the share-local shift is off by at most one
    for c_ in captured:
        be.tensor(c_).zero_()
    g.replay()
    torch.cuda.synchronize()
    for j in range(2):
        assert np.array_equal(be.down(captured[j]), want[j]), f"share {j} of the replay differs from the eager pass"
