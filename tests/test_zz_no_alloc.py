"""The entry point that allocates nothing (mpcx_ring_gemm_u64_noalloc: limb planes in a static device array) against
mpcx_ring_gemm_u64 and the CPU oracle: two calls in a row, the second with a smaller input, then the size limit."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_noalloc_ring_gemm_matches_the_allocating_entry_point(orc):
    from backends import GpuBackend
    from mpc_amd import MpcxError, load_library
    gpu = GpuBackend()
    ctx = gpu.ctx
    rng = np.random.default_rng(77)
    limit = load_library().mpcx_ring_gemm_noalloc_limit
    limit.restype = ctypes.c_uint64
    assert limit() == 16 << 20
    # the first call fills the static planes with a large product, the second reuses them for a smaller, ragged one
    for r, l, c, la, lb, lc, acc in ((1000, 1024, 1024, 0, 0, 0, 0), (130, 9000, 66, 1, 0, 1, 1), (5, 4, 3, 0, 1, 0, 0)):
        A = rng.integers(-2 ** 63, 2 ** 63, r * l, dtype=np.int64)
        B = rng.integers(-2 ** 63, 2 ** 63, l * c, dtype=np.int64)
        C0 = rng.integers(-2 ** 63, 2 ** 63, r * c, dtype=np.int64)
        dA, dB = ctx.to_device(A), ctx.to_device(B)
        out = {}
        for name in ("ring_gemm_u64_noalloc", "ring_gemm_u64"):
            dC = ctx.to_device(C0)
            ctx.call(name, dA, dB, dC, r, l, c, la, lb, lc, acc)
            ctx.sync()
            out[name] = dC.to_host()
        assert np.array_equal(out["ring_gemm_u64_noalloc"], out["ring_gemm_u64"]), (r, l, c)
        assert np.array_equal(out["ring_gemm_u64"], orc.ring_gemm(A, B, r, l, c, la, lb, lc, C=C0 if acc else None)), (r, l, c)
    # the size query: the halving of long contractions (l = 9000 -> chunks of 4544), and the limit's edge
    nbytes = load_library().mpcx_ring_gemm_noalloc_bytes
    nbytes.restype = ctypes.c_uint64
    U = ctypes.c_uint64
    assert nbytes(U(130), U(9000), U(66)) == 8 * (256 + 128) * 4544
    assert nbytes(U(1024), U(1024), U(1024)) == 16 << 20 and nbytes(U(1025), U(1024), U(1024)) > 16 << 20
    one = ctx.to_device(np.zeros(1, np.int64))
    with pytest.raises(MpcxError, match="static scratch holds"):
        ctx.call("ring_gemm_u64_noalloc", one, one, one, 1025, 1024, 1024, 0, 0, 0, 0)
    with pytest.raises(MpcxError, match="bad shape"):
        ctx.call("ring_gemm_u64_noalloc", one, one, one, 0, 1, 1, 0, 0, 0, 0)


def test_noalloc_ring_gemm_leaves_device_memory_untouched():
    """On a context that has never run a product (its workspace is empty), a call of the no-alloc entry point at a size
    that would make the allocating one grow the workspace by 16 MiB changes neither the free device memory nor the
    context's workspace; the allocating entry point, called next as the control, does."""
    import torch
    from mpc_amd import Context
    ctx = Context(0)
    r = l = c = 1024
    rng = np.random.default_rng(5)
    dA = ctx.to_device(rng.integers(-2 ** 63, 2 ** 63, r * l, dtype=np.int64))
    dB = ctx.to_device(rng.integers(-2 ** 63, 2 ** 63, l * c, dtype=np.int64))
    dC, dD = ctx.empty(np.int64, r * c), ctx.empty(np.int64, r * c)
    ctx.sync()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for _ in range(3):
        ctx.call("ring_gemm_u64_noalloc", dA, dB, dC, r, l, c, 0, 0, 0, 0)
    ctx.sync()
    assert torch.cuda.mem_get_info(0)[0] == free0, "the no-alloc entry point changed the free device memory"
    ctx.call("ring_gemm_u64", dA, dB, dD, r, l, c, 0, 0, 0, 0)  # the control: this one takes a workspace
    ctx.sync()
    assert torch.cuda.mem_get_info(0)[0] <= free0 - (16 << 20)
    assert np.array_equal(dC.to_host(), dD.to_host())
    ctx.close()
