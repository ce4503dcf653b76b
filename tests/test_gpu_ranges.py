"""Element-range sharding on the GPU (one range per GPU at --gpus N): the HIP segments run on ranges
(n_total, i0, n) of a vector must reproduce, message by message and share by share, the CPU oracle's run over the
whole vector."""
import numpy as np
import pytest

import protocols as pr
from oracle_lib import prg_pops
from range_helpers import aby3_relu, run_in_ranges, same_logs, snn_msb

pytestmark = pytest.mark.gpu

RANGES = [((0, 1700), (1700, 1), (1701, 3299)), ((0, 2500), (2500, 2500))]


@pytest.fixture(scope="module")
def gpu():
    from backends import GpuBackend
    return GpuBackend()


@pytest.mark.parametrize("ranges", RANGES)
def test_aby3_relu_chain_in_ranges(gpu, orc, ranges):
    x = prg_pops(21, 0, 5000)
    wo = pr.World(orc)
    want = aby3_relu(wo, x)
    got, log = run_in_ranges(gpu, aby3_relu, x, ranges, halves=4)
    same_logs(log, wo.log)
    for j in range(3):
        assert np.array_equal(got[j], want[j]), f"party {j}"
    n = x.size
    prod = (want[0][:n].view(np.uint64) + want[1][:n].view(np.uint64) + want[2][:n].view(np.uint64)).view(np.int64)
    assert np.array_equal(prod, np.where(x < 0, x, 0))
    relu = sum(want[j][2 * n:3 * n].view(np.uint64) for j in range(3)).view(np.int64)
    assert np.array_equal(relu, np.maximum(x, 0))


@pytest.mark.parametrize("ranges", RANGES)
def test_snn_msb_in_ranges(gpu, orc, ranges):
    a = prg_pops(22, 0, 5000) >> 1
    wo = pr.World(orc)
    want = snn_msb(wo, a)
    got, log = run_in_ranges(gpu, snn_msb, a, ranges)
    same_logs(log, wo.log)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert np.array_equal((got[0].view(np.uint64) + got[1].view(np.uint64)).view(np.int64), (a < 0).astype(np.int64))


def test_aby3_vecmul_and_trunc_in_ranges(gpu, orc):
    n = 3000
    x, y = prg_pops(23, 0, 2 * n).reshape(2, n)
    x, y = x >> 40, y >> 40

    def chain(w, xy):
        m = xy.size // 2
        Xs, Ys = pr.aby3_share(w, xy[:m].copy(), m), pr.aby3_share(w, xy[m:].copy(), m)
        Z = pr.aby3_trunc(w, pr.aby3_vecmul(w, Xs, Ys, m), 8)
        return [np.concatenate([np.asarray(z[0]), np.asarray(z[1])]) for z in Z]

    wo = pr.World(orc)
    want = chain(wo, np.concatenate([x, y]))
    outs = []
    ranges = ((0, 1000), (1000, 2000))
    for i0, m in ranges:
        w = pr.World(gpu, n_total=n, i0=i0)
        outs.append(chain(w, np.concatenate([x[i0:i0 + m], y[i0:i0 + m]])))
    for j in range(3):
        got = np.concatenate([o[j].reshape(2, m) for o, (_, m) in zip(outs, ranges)], axis=1).ravel()
        assert np.array_equal(got, want[j]), f"party {j}"
