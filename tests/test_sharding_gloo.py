"""Multi-rank path on CPU: two gloo ranks each compute their row-block / element range of a party's
local segment (with the CPU oracle standing in for the device kernels) and gather; the result must equal
the single-rank computation.  This exercises the sharding arithmetic bench.py uses at --gpus N:
row offsets into the row-major mask streams, (n_total, i0, n) element ranges and the all-gather layout."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from backends import OracleBackend
    from mpc_amd.sharding import aby3_matmul_pop_plan, matmul_row_block, split_range
    from oracle_lib import KEY01, KEY20, prg_pops
    orc = OracleBackend()
    r, l, c = 96, 40, 72
    rng = np.random.default_rng(9)
    X0, X1, Y0, Y1 = [rng.integers(-2 ** 63, 2 ** 63, k, dtype=np.int64) for k in (r * l, r * l, l * c, l * c)]
    plan = aby3_matmul_pop_plan(r, l, c)
    row0, rows = matmul_row_block(r, world, rank)
    pop = plan["matmul"] + row0 * c
    blk = orc.aby3_matmul_local(X0[row0 * l:(row0 + rows) * l], X1[row0 * l:(row0 + rows) * l], Y0, Y1, rows, l, c, 0,
                                KEY01, pop, KEY20, pop)
    parts = [torch.empty(rows * c, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(parts, torch.from_numpy(blk))
    Z0 = torch.cat(parts).numpy()
    whole = orc.aby3_matmul_local(X0, X1, Y0, Y1, r, l, c, 0, KEY01, plan["matmul"], KEY20, plan["matmul"])
    ok = np.array_equal(Z0, whole)

    # element ranges: privateCompare (party role) over a sharded vector
    n = 400
    w = prg_pops(22, 0, 67 * n).reshape(n, 67)
    x, rr = w[:, 0].copy(), w[:, 1].copy()
    rr[2::5] = -1
    beta = (w[:, 2].view(np.uint64) % 2).astype(np.int8)
    s0 = np.ascontiguousarray(((w[:, 3:].view(np.uint64) & 0xFFFF) % 67).astype(np.uint16)).ravel()
    i0, m = split_range(n, world, rank)
    d = orc.snn_pc_party(0, s0[64 * i0:64 * (i0 + m)], beta[i0:i0 + m], rr[i0:i0 + m], n, i0, KEY01)
    dparts = [torch.empty(16 * m, dtype=torch.int64) for _ in range(world)]  # 128 bytes per element
    dist.all_gather(dparts, torch.from_numpy(d.view(np.int64)))
    dall = torch.cat(dparts).numpy().view(np.uint16)
    ok = ok and np.array_equal(dall, orc.snn_pc_party(0, s0, beta, rr, n, 0, KEY01))

    # strong scaling, the shape of the 16384 x 16384 configuration at a reduced row count: c = 16384 columns, the row
    # block's mask words start row0 * c words into both pop streams (ABY3Engine.cpp:190-197: row-major order), and one of
    # the two streams is read beyond 2^32 words
    r5, l5, c5 = 8, 4, 16384
    A0, A1, B0, B1 = [rng.integers(-2 ** 63, 2 ** 63, k, dtype=np.int64) for k in (r5 * l5, r5 * l5, l5 * c5, l5 * c5)]
    far, near = (1 << 32) + 12345, aby3_matmul_pop_plan(r5, l5, c5)["matmul"]
    row0, rows = matmul_row_block(r5, world, rank)
    blk = orc.aby3_matmul_local(A0[row0 * l5:(row0 + rows) * l5], A1[row0 * l5:(row0 + rows) * l5], B0, B1, rows, l5, c5, 0,
                                KEY01, far + row0 * c5, KEY20, near + row0 * c5)
    parts = [torch.empty(rows * c5, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(parts, torch.from_numpy(blk))
    ok = ok and np.array_equal(torch.cat(parts).numpy(), orc.aby3_matmul_local(A0, A1, B0, B1, r5, l5, c5, 0, KEY01, far, KEY20, near))
    # element ranges whose pop offsets straddle 2^32 words: share (one word per element and stream)
    ns, edge = 400, (1 << 32) - 150
    xs = rng.integers(-2 ** 63, 2 ** 63, ns, dtype=np.int64)
    i0, m = split_range(ns, world, rank)
    sh = orc.aby3_share_local(xs[i0:i0 + m], m, KEY01, edge + i0, KEY20, 77 + i0)
    sparts = [torch.empty(m, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sparts, torch.from_numpy(sh))
    ok = ok and np.array_equal(torch.cat(sparts).numpy(), orc.aby3_share_local(xs, ns, KEY01, edge, KEY20, 77))
    open(os.path.join(out_dir, f"rank{rank}.ok"), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_row_blocks_and_element_ranges(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        assert open(tmp_path / f"rank{rank}.ok").read() == "1"


def test_split_range_covers_everything():
    from mpc_amd.sharding import split_range
    for total in (0, 1, 7, 8, 4096, 12345):
        for parts in (1, 2, 3, 8):
            got = [split_range(total, parts, p) for p in range(parts)]
            assert got[0][0] == 0 and sum(c for _, c in got) == total
            for (f0, c0), (f1, _) in zip(got, got[1:]):
                assert f0 + c0 == f1


def test_snn_msb_in_element_ranges(orc):
    """The SecureNN msb protocol run range by range (n_total, i0, n) gives the whole vector's messages and shares
    (the CPU restatement takes the same range arguments as the device segments)."""
    import protocols as pr
    from oracle_lib import prg_pops
    from range_helpers import run_in_ranges, same_logs, snn_msb
    a = prg_pops(22, 0, 1200) >> 1
    wo = pr.World(orc)
    want = snn_msb(wo, a)
    got, log = run_in_ranges(orc, snn_msb, a, ((0, 500), (500, 1), (501, 699)))
    same_logs(log, wo.log)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
