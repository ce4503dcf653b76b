#!/usr/bin/env python3
"""Benchmark of the ABY3 share-matmul hot path on MI355X.

Workload (BASELINE.json configs[1]): ABY3 4096 x 4096 Z_2^64 share matmul + reconstruct.  One step =
for ALL THREE parties, run on this GPU one after the other with every input already resident in HBM:
share_matrix(X), share_matrix(Y) (PRG regeneration + mask), matrix_mul's local part (int8-limb MFMA ring
GEMM of [X0|X1]*[Y0+Y1;Y0] + fused PRG mask) and reveal_matrix's sum.  Party-to-party messages are
device buffers handed over by pointer (the messenger round is not part of the hot path).

--gpus N (launched by torch.distributed.run, one rank per GPU): weak scaling over matmul row-blocks --
X has 4096*N rows, rank g owns rows [4096 g, 4096 (g+1)), Y is replicated, and the row-blocks of each
party's Z0 (the message to the previous party) are gathered with one RCCL gather per party to rank 0, the rank that
faces the messenger (--gather all: an all-gather), which stays in flight while the following products run and is
completed before the timed region ends.  The
privateCompare / msb / ReLU items run on per-rank element ranges (n_total, i0, n) and are merged over ranks.

value = ring operations of the reference's formulation (3 products, 2 ops per multiply-add:
3 parties x 3*2*r*l*c) per second, whole job.  Extra keys report the privateCompare / MSB element rates,
the roofline of the dominant kernel and the CPU oracle timed on the host cores."""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mpc_amd.sharding import aby3_matmul_pop_plan, matmul_row_block, split_range  # noqa: E402

KEY01, KEY12, KEY20, KEYS = 0xAB01, 0xBC12, 0xCA20, 0xF0
INT8_PEAK_TOPS = 5033.0  # 256 CU x 4 SIMD x 2048 int8 op/clk x 2.4 GHz (2x the 2.5 PF bf16 dense peak)
HBM_PEAK_GBS = 8000.0


def pair_seed(a, b):
    return {frozenset((0, 1)): KEY01, frozenset((1, 2)): KEY12, frozenset((0, 2)): KEY20}[frozenset((a, b))]


PMC_SUMMARY = "profiles/r3_pmc_summary.json"
PMC_BITWISE = "profiles/r3_pmc_bitwise.json"


def measured_traffic(n, world, kernel="limb_gemm_kernel"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE doubled
    as gfx950 requires, + WRITE_SIZE); they were taken on the default single-GPU workload only."""
    path = os.path.join(ROOT, PMC_SUMMARY)
    if n != 4096 or world != 1 or not os.path.exists(path):
        return None
    d = json.load(open(path)).get(kernel)
    if not d or "derived" not in d:
        return None
    return d["derived"]["hbm_read_bytes_per_launch"] + d["derived"]["hbm_write_bytes_per_launch"]


class TableBuilder:
    """PRG checkpoint tables are a serial host walk of a stream whose key is a compile-time constant: on a node with
    several ranks one rank walks and the others load what it left in the library's on-disk cache (MPCX_CKPT_CACHE)."""

    def __init__(self, ctx, dist, rank, world):
        self.ctx, self.dist, self.rank, self.shared = ctx, dist, rank, dist is not None and world > 1
        self.made_dir = None
        if self.shared:
            if "MPCX_CKPT_CACHE" not in os.environ:  # a directory of this job's own, removed when the job ends
                import tempfile
                box = [tempfile.mkdtemp(prefix="mpcx_ckpt_") if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                self.made_dir = box[0]
                os.environ["MPCX_CKPT_CACHE"] = self.made_dir
            os.makedirs(os.environ["MPCX_CKPT_CACHE"], exist_ok=True)

    def cleanup(self):
        if self.made_dir and self.rank == 0:
            import shutil
            shutil.rmtree(self.made_dir, ignore_errors=True)

    def build(self, specs):
        """specs: (seed, n_pops, n_buff) triples; the streams are walked on as many host threads."""
        def walk():
            th = [threading.Thread(target=self.ctx.call, args=("prg_ckpt_build", s, 0, p, b)) for s, p, b in specs]
            [t.start() for t in th]
            [t.join() for t in th]
        if not self.shared:
            return walk()
        if self.rank == 0:
            walk()
        self.dist.barrier()
        if self.rank != 0:
            walk()  # finds the tables on disk


def prof_read(ctx, name):
    ms, cnt = ctypes.c_double(), ctypes.c_uint64()
    ctx._call("mpcx_profile_read", name.encode(), ctypes.byref(ms), ctypes.byref(cnt))
    return ms.value, cnt.value


class Aby3MatmulJob:
    def __init__(self, ctx, torch, dev, n, rank, world, host_gather=False, rows=None, gather=None, gather_to="root",
                 tables=None):
        self.ctx, self.torch, self.n, self.rank, self.world = ctx, torch, n, rank, world
        tables = tables or TableBuilder(ctx, None, rank, 1)
        self.gather = world > 1 if gather is None else gather  # the collective also runs in a 1-rank process group
        self.host_gather = host_gather
        self.gather_to = gather_to  # "root": only rank 0 (the rank that faces the messenger) receives; "all": all-gather
        self.rows, self.l, self.c = rows or n, n, n
        self.rows_total = self.rows * world
        self.row0, rows_here = matmul_row_block(self.rows_total, world, rank)
        assert rows_here == self.rows
        self.plan = aby3_matmul_pop_plan(self.rows_total, self.l, self.c)
        i64 = torch.int64
        e = lambda k: torch.empty(k, dtype=i64, device=dev)
        # pop-stream checkpoints for everything a step consumes (fresh engines every step) and for the inputs
        pops = self.plan["total"]
        tables.build([(k, pops, 0) for k in (KEY01, KEY12, KEY20)] + [(20, self.rows_total * self.l + self.l * self.c, 0)])
        # plaintext inputs: X then Y row-major from PRgenerator(20,0).pop_int64()
        self.X, self.Y = e(self.rows * self.l), e(self.l * self.c)
        ctx.call("prg_fill_pop", 20, 0, self.row0 * self.l, self.rows * self.l, self.X)
        ctx.call("prg_fill_pop", 20, 0, self.rows_total * self.l, self.l * self.c, self.Y)
        self.RX = [e(self.rows * self.l) for _ in range(3)]
        self.RY = [e(self.l * self.c) for _ in range(3)]
        self.Z0 = [e(self.rows * self.c) for _ in range(3)]
        self.Zfull = ([e(self.rows_total * self.c) for _ in range(3)]
                      if self.gather and (rank == 0 or gather_to == "all") else None)
        self.Zplain = [e(self.rows * self.c) for _ in range(3)]
        self.pending = [None, None, None]  # all-gathers in flight, per party
        ctx.sync()

    def step(self):
        ctx, n2y = self.ctx, self.l * self.c
        n2x, pm = self.plan["share_y"], self.plan["matmul"]
        for j in range(3):
            nx, pv = (j + 1) % 3, (j + 2) % 3
            sn, sp = pair_seed(j, nx), pair_seed(j, pv)
            off = self.row0 * self.l
            ctx.call("aby3_share_local", self.X if j == 0 else None, self.rows * self.l, sn, off, sp, off, self.RX[j])
            ctx.call("aby3_share_local", self.Y if j == 0 else None, n2y, sn, n2x, sp, n2x, self.RY[j])
        for j in range(3):
            nx, pv = (j + 1) % 3, (j + 2) % 3
            sn, sp = pair_seed(j, nx), pair_seed(j, pv)
            if self.pending[j] is not None:  # the previous step's message of this party still reads Z0[j]
                self.pending[j].wait()
                self.pending[j] = None
            ctx.call("aby3_matmul_local_rows", self.RX[j], self.RX[nx], self.RY[j], self.RY[nx], self.rows, self.l,
                     self.c, self.row0, sn, pm, sp, pm, self.Z0[j])
            if self.gather:
                # the message party j hands to the messenger is all row-blocks of its Z0: one gather per party,
                # started as soon as the block exists.  It travels over xGMI while the following products run -- of
                # this step and, for the last party, of the next one (the steps are independent jobs); it is waited
                # for before its source block is overwritten, and drain() completes the last ones.
                import torch.distributed as dist
                root = self.gather_to == "root"
                if self.host_gather:  # rehearsal without RCCL (gloo): through host memory
                    parts = [self.torch.empty(self.Z0[j].numel(), dtype=self.torch.int64) for _ in range(self.world)]
                    if root:
                        dist.gather(self.Z0[j].cpu(), parts if self.rank == 0 else None, dst=0)
                    else:
                        dist.all_gather(parts, self.Z0[j].cpu())
                    if self.Zfull is not None:
                        self.Zfull[j].copy_(self.torch.cat(parts))
                elif root:
                    outs = list(self.Zfull[j].chunk(self.world)) if self.rank == 0 else None
                    self.pending[j] = dist.gather(self.Z0[j], outs, dst=0, async_op=True)
                else:
                    self.pending[j] = dist.all_gather_into_tensor(self.Zfull[j], self.Z0[j], async_op=True)
        for j in range(3):
            ctx.call("aby3_reveal3", self.Z0[(j + 2) % 3], self.Z0[j], self.Z0[(j + 1) % 3], self.Zplain[j],
                     self.rows * self.c)

    def drain(self):
        """Completes the messages still in flight (inside the timed region: a job is done when its messages are)."""
        for j in range(3):
            if self.pending[j] is not None:
                self.pending[j].wait()
                self.pending[j] = None

    def ring_ops_per_step(self):
        return 3 * 3 * 2 * self.rows_total * self.l * self.c

    def shape_only(self):
        """The sizes without the device buffers (for the report after the buffers have been released)."""
        o = object.__new__(Aby3MatmulJob)
        o.rows, o.l, o.c, o.rows_total = self.rows, self.l, self.c, self.rows_total
        return o

    def verify(self):
        """All parties reveal the same Z, and sampled entries equal the exact big-integer dot products."""
        t = self.torch
        assert t.equal(self.Zplain[0], self.Zplain[1]) and t.equal(self.Zplain[1], self.Zplain[2])
        if self.gather:
            # every rank's block of every gathered message: its wrap-around sum travels separately and is compared with
            # the sum of the block that arrived; a rank's own block is compared word for word
            import torch.distributed as dist
            mine = [int(self.Z0[j].sum().item()) for j in range(3)]
            sums = [None] * self.world
            dist.all_gather_object(sums, mine)
            if self.Zfull is not None:
                nb = self.rows * self.c
                for j in range(3):
                    lo = self.row0 * self.c
                    assert t.equal(self.Zfull[j][lo:lo + nb], self.Z0[j]), "gathered message differs"
                    for g in range(self.world):
                        assert int(self.Zfull[j][g * nb:(g + 1) * nb].sum().item()) == sums[g][j], \
                            f"block of rank {g} in the message of party {j} differs"
        X = self.X.view(self.rows, self.l).cpu().numpy().view(np.uint64)
        Y = self.Y.view(self.l, self.c).cpu().numpy().view(np.uint64)
        Z = self.Zplain[0].view(self.rows, self.c).cpu().numpy().view(np.uint64)
        rng = np.random.default_rng(1)
        for _ in range(24):
            i, k = int(rng.integers(self.rows)), int(rng.integers(self.c))
            want = sum(int(a) * int(b) for a, b in zip(X[i], Y[:, k])) % (1 << 64)
            assert int(Z[i, k]) == want, f"Z[{i},{k}] wrong"
        # every entry at once: Z v == X (Y v) over Z_2^64 for random vectors v (numpy uint64 arithmetic wraps)
        with np.errstate(over="ignore"):
            for _ in range(4):
                v = rng.integers(0, 1 << 64, self.c, dtype=np.uint64) | np.uint64(1)
                assert np.array_equal(Z @ v, X @ (Y @ v)), "Z v != X (Y v)"


def cpu_baseline(n=2048):
    """The CPU oracle (a port of the reference algorithm) on the same workload at a bounded size: three
    party threads, each single-threaded, as the reference runs three single-threaded processes."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))  # the oracle backend is test infrastructure: cpu_baseline leg only
    from backends import OracleBackend
    from oracle_lib import prg_pops
    orc = OracleBackend()
    wd = prg_pops(20, 0, 2 * n * n)
    X, Y = wd[: n * n].copy(), wd[n * n:].copy()
    R = {}

    def share(j):
        nx, pv = (j + 1) % 3, (j + 2) % 3
        sn, sp = pair_seed(j, nx), pair_seed(j, pv)
        R[("x", j)] = orc.aby3_share_local(X if j == 0 else None, n * n, sn, 0, sp, 0)
        R[("y", j)] = orc.aby3_share_local(Y if j == 0 else None, n * n, sn, n * n, sp, n * n)

    def mul(j):
        nx, pv = (j + 1) % 3, (j + 2) % 3
        sn, sp = pair_seed(j, nx), pair_seed(j, pv)
        R[("z", j)] = orc.aby3_matmul_local(R[("x", j)], R[("x", nx)], R[("y", j)], R[("y", nx)], n, n, n, 0, sn,
                                            2 * n * n, sp, 2 * n * n)

    def rev(j):
        R[("p", j)] = orc.aby3_reveal3(R[("z", (j + 2) % 3)], R[("z", j)], R[("z", (j + 1) % 3)])

    def phase(fn):
        th = [threading.Thread(target=fn, args=(j,)) for j in range(3)]
        [t.start() for t in th]
        [t.join() for t in th]

    orc.prg_fill_pop(KEY01, 0, 3 * n * n), orc.prg_fill_pop(KEY12, 0, 3 * n * n), orc.prg_fill_pop(KEY20, 0, 3 * n * n)
    t0 = time.perf_counter()
    phase(share)
    phase(mul)
    phase(rev)
    dt = time.perf_counter() - t0
    cpu = "unknown"
    try:
        cpu = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {"value": 3 * 6 * n ** 3 / dt / 1e9, "unit": "Gop/s", "cores": 3, "kind": "port",
            "sample": f"ABY3 {n}x{n} share_matrix x2 + matrix_mul + reveal_matrix, 3 party threads, {dt:.2f} s wall",
            "host": {"cpu": cpu, "logical_cpus": os.cpu_count()}}


def cpu_bitwise(log2n=18):
    """The CPU oracle on the privateCompare / msb / ReLU items at a bounded size (they are O(n)): one thread plays
    every role in turn, as bitwise_rates does on the GPU.  Inputs as there (privateCompare: r = -1 on every fifth
    element; msb: pop / 2; ReLU: pops)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))  # the oracle backend is test infrastructure: cpu_baseline leg only
    import mpc_amd.protocols as pr
    from backends import OracleBackend
    orc = OracleBackend()
    n = 1 << log2n
    rng = np.random.default_rng(3)
    out = {"n": n, "cores": 1, "kind": "port"}

    def timed(fn):
        t0 = time.perf_counter()
        r = fn()
        return time.perf_counter() - t0, r

    xb = rng.integers(0, 67, 64 * n, dtype=np.uint16)
    beta = rng.integers(0, 2, n, dtype=np.int8)
    r = rng.integers(-2 ** 62, 2 ** 62, n, dtype=np.int64)
    r[2::5] = -1
    d0 = orc.snn_pc_party(0, xb, beta, r, n, 0, KEY01)  # first call: the generator buffers are extended to 128 n words
    dt, d0 = timed(lambda: orc.snn_pc_party(0, xb, beta, r, n, 0, KEY01))
    out["pc_party_elems_per_s"] = n / dt
    d1 = orc.snn_pc_party(1, xb, beta, r, n, 0, KEY01)
    dt, _ = timed(lambda: orc.snn_pc_helper(d0, d1))
    out["pc_helper_elems_per_s"] = n / dt

    a = orc.prg_fill_pop(22, 0, n) // 2

    def snn_msb():
        w = pr.World(orc, record=False)
        return pr.snn_msb(w, pr.snn_share(w, a, n)[:2])

    snn_msb()
    dt, B = timed(snn_msb)
    assert np.array_equal(B[0] + B[1], (a < 0).astype(np.int64))
    out["snn_msb_elems_per_s"] = n / dt

    x = orc.prg_fill_pop(21, 0, n)

    def aby3_relu():
        w = pr.World(orc, record=False)
        Xs = pr.aby3_share(w, x, n)
        Gs, Ps = pr.aby3_add_prepare(w, Xs)
        C = pr.aby3_bit_inject(w, Xs, pr.aby3_msb(w, Gs, Ps))
        return [orc.ring_sub(Xs[j][0], C[j][0]) for j in range(3)]

    dt, R = timed(aby3_relu)
    assert np.array_equal(R[0] + R[1] + R[2], np.maximum(x, 0))
    out["aby3_relu_elems_per_s"] = n / dt
    return out


def _timed(torch, fn, reps=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, r


def private_compare_inputs(ctx, torch, dev, i0, n):
    """privateCompare inputs for elements [i0, i0 + n) as test/test_securenn.cpp:160-175 draws them from
    PRgenerator(22,0), 67 pops per element: x, r (forced to -1 when i % 5 == 2), beta = the low bit of a pop, then for
    every bit position k the share xs0[k] = (uint16)pop % 67 of party 0; party 1's share is
    (bit_k(x) + 67 - xs0[k]) % 67.  Returns (xbits0, xbits1, beta, r) on the device."""
    xb = torch.empty(64 * n, dtype=torch.int16, device=dev)
    xb1 = torch.empty(64 * n, dtype=torch.int16, device=dev)
    beta = torch.empty(n, dtype=torch.int8, device=dev)
    r = torch.empty(n, dtype=torch.int64, device=dev)
    step = 1 << 18
    kk = torch.arange(64, device=dev)
    for a in range(0, n, step):
        m = min(step, n - a)
        w = torch.empty(67 * m, dtype=torch.int64, device=dev)
        ctx.call("prg_fill_pop", 22, 0, 67 * (i0 + a), 67 * m, w)
        w = w.view(m, 67)
        x = w[:, 0]
        r[a:a + m] = w[:, 1]
        beta[a:a + m] = (w[:, 2] & 1).to(torch.int8)
        s0 = (w[:, 3:] & 0xFFFF) % 67
        xb[64 * a:64 * (a + m)] = s0.to(torch.int16).reshape(-1)
        xb1[64 * a:64 * (a + m)] = ((((x.unsqueeze(1) >> kk) & 1) + 67 - s0) % 67).to(torch.int16).reshape(-1)
    idx = torch.arange(i0, i0 + n, device=dev)
    r[idx % 5 == 2] = -1  # test/test_securenn.cpp:165-166
    return xb, xb1, beta, r


def bitwise_table_specs(log2n, world):
    """(seed, n_pops, n_buff) of every generator stream bitwise_rates reads: the SecureNN buffers of a 2^log2n-element
    vector, its inputs from PRgenerator(22,0), and the pop streams of the ABY3 chain on world * 2^log2n elements (share:
    one word per element and stream, bit_inject: up to six more)."""
    nt = 1 << log2n
    return ([(KEY01, 7 * nt * world, 128 * nt), (KEYS, 0, 67 * nt + 64), (22, 67 * nt, 0), (21, nt * world, 0)] +
            [(k, 7 * nt * world, 0) for k in (KEY12, KEY20)])


def bitwise_rates(ctx, torch, dev, log2n=24, rank=0, world=1, tables=None):
    """Secondary metrics, elements / s with every buffer resident in HBM (this rank's element range; merged over
    the ranks by merge_bitwise):
      * privateCompare, party role and helper role, as kernels (config 3's n = 2^24);
      * the whole SecureNN msb protocol (shareConvert + msb_L_1, all three roles played on this GPU, messages
        handed over as device pointers) and its per-segment device time;
      * the ABY3 ReLU chain add_prepare -> msb -> bit_inject -> subtract, all three parties on this GPU.
    "cold" regenerates the int_buff-derived random bytes inside the call; "warm" reads them from the device
    cache, which is what the reference does after its constructor has filled int_buff.

    With several ranks every rank works on a contiguous element range (n_total, i0, n) of one vector.  The SecureNN
    items split a vector of 2^log2n elements (its buffer offsets reach 256 * n_total words of a serial generator,
    and walking those once per rank is set-up time that would dominate this script if n_total grew with the ranks);
    the ABY3 chain, which only pops, gives every rank 2^log2n elements of a vector of world * 2^log2n."""
    import mpc_amd.protocols as pr
    from mpc_amd.backends import ResidentBackend
    be = ResidentBackend(ctx, torch, dev)
    GiB = 1 << 30
    nt = 1 << log2n                      # SecureNN: the whole vector
    i0, n = split_range(nt, world, rank)  # ... and this rank's range of it
    out = {"n": nt, "n_this_rank": n}
    counts = {}                           # elements behind every *_elems_per_s entry (for the merge over ranks)

    def rate(key, elems, dt):
        out[key] = elems / dt
        counts[key] = elems

    # ---- privateCompare kernels
    t0 = time.perf_counter()
    # the generators are walked on one host thread each (every stream has its own lock); with several ranks on the node
    # one rank walks and the others load the tables it leaves on disk
    # (main() has built them already when it shares them between ranks: no collective may run in here, where a rank
    # that fails must still reach the gather of the results)
    TableBuilder(ctx, None, rank, 1).build(bitwise_table_specs(log2n, world))
    out["ckpt_build_s"] = round(time.perf_counter() - t0, 2)
    # Inputs as test/test_securenn.cpp:160-175 draws them from PRgenerator(22,0), 67 pops per element: x, r (forced to
    # -1 when i % 5 == 2), beta = the low bit of a pop, then for every bit position k the share xs0[k] = (uint16)pop % 67
    # of party 0; party 1's share (bit_k(x) + 67 - xs0[k]) % 67 is what the party role of party 1 would be handed.
    xb, xb1, beta, r = private_compare_inputs(ctx, torch, dev, i0, n)
    d0 = torch.empty(64 * n, dtype=torch.int16, device=dev)
    d1 = torch.empty(64 * n, dtype=torch.int16, device=dev)
    b2 = torch.empty(n, dtype=torch.int8, device=dev)
    for label, cap in (("cold", 0), ("warm", 8 * GiB)):
        ctx.call("set_rand_cache", cap)
        ctx.call("snn_pc_party", 0, xb, beta, r, nt, i0, n, KEY01, d0)
        ctx.call("snn_pc_party", 1, xb1, beta, r, nt, i0, n, KEY01, d1)
        ctx.call("profile", 1)
        dt, _ = _timed(torch, lambda: ctx.call("snn_pc_party", 0, xb, beta, r, nt, i0, n, KEY01, d0), reps=3)
        ms, cnt = prof_read(ctx, "pc_party")
        msp, _ = prof_read(ctx, "prg_fill")
        ctx.call("profile", 0)
        rate(f"pc_party_{label}_elems_per_s", n, dt)
        out[f"pc_party_{label}_kernel_ms"] = round(ms / cnt, 3)
        out[f"pc_party_{label}_prg_ms"] = round(msp / 3, 3)
    dt, _ = _timed(torch, lambda: ctx.call("snn_pc_helper", d0, d1, n, b2), reps=3)
    rate("pc_helper_elems_per_s", n, dt)
    # the recipe's own check (test/test_securenn.cpp:201-209): beta xor beta' = (x > r) unsigned, on the first chunk
    m = min(n, 1 << 18)
    w = torch.empty(67 * m, dtype=torch.int64, device=dev)
    ctx.call("prg_fill_pop", 22, 0, 67 * i0, 67 * m, w)
    x = w.view(m, 67)[:, 0]
    top = torch.iinfo(torch.int64).min
    assert torch.equal((beta[:m] ^ b2[:m]).bool(), (x ^ top) > (r[:m] ^ top)), "privateCompare: beta ^ beta' != (x > r)"
    del xb, xb1, d0, d1, b2, beta, r, w, x

    # ---- SecureNN msb, whole protocol (test/test_securenn.cpp:283-308: a = pop_int64() / 2)
    a = torch.empty(n, dtype=torch.int64, device=dev)
    ctx.call("prg_fill_pop", 22, 0, i0, n, a)
    a = torch.div(a, 2, rounding_mode="trunc")
    aD = ctx.wrap(a, np.int64)

    def snn_msb():
        w = pr.World(be, record=False, n_total=nt, i0=i0)
        As = pr.snn_share(w, aD, n)
        return pr.snn_msb(w, As[:2])

    for label, cap in (("cold", 0), ("warm", 8 * GiB)):
        ctx.call("set_rand_cache", cap)
        snn_msb()
        be.events = []
        dt, B = _timed(torch, snn_msb)
        seg = be.segment_ms()
        be.events = None
        rate(f"snn_msb_{label}_elems_per_s", n, dt)
        out[f"snn_msb_{label}_ms"] = round(dt * 1e3, 3)
        out[f"snn_msb_{label}_segments_ms"] = {k: round(v, 3) for k, v in sorted(seg.items(), key=lambda kv: -kv[1])}
    got = torch.from_numpy(be.down(B[0])).to(dev) + torch.from_numpy(be.down(B[1])).to(dev)
    assert torch.equal(got, (a < 0).to(torch.int64)), "SecureNN msb: revealed bits differ from sign(a)"
    del B, got

    # ---- ABY3 ReLU chain (test/test_ABY3.cpp:196-214, benchmark/benchmark1.cpp:69-74)
    n = 1 << log2n  # ABY3: every rank's range
    nt, i0 = n * world, n * rank
    out["aby3_relu_n"] = nt
    x = torch.empty(n, dtype=torch.int64, device=dev)
    ctx.call("prg_fill_pop", 21, 0, i0, n, x)
    xD = ctx.wrap(x, np.int64)

    def aby3_relu():
        w = pr.World(be, record=False, n_total=nt, i0=i0)
        Xs = pr.aby3_share(w, xD, n)
        Gs, Ps = pr.aby3_add_prepare(w, Xs)
        Zb = pr.aby3_msb(w, Gs, Ps)
        R = pr.aby3_bit_inject(w, Xs, Zb, minuend=Xs)  # both result shares of every party: x - bit_inject(x, msb)
        return [R[j][0] for j in range(3)]

    aby3_relu()
    be.events = []
    dt, R = _timed(torch, aby3_relu)
    seg = be.segment_ms()
    be.events = None
    rate("aby3_relu_elems_per_s", n, dt)
    out["aby3_relu_ms"] = round(dt * 1e3, 3)
    out["aby3_relu_segments_ms"] = {k: round(v, 3) for k, v in sorted(seg.items(), key=lambda kv: -kv[1])}
    got = sum(torch.from_numpy(be.down(R[j])).to(dev) for j in range(3))
    assert torch.equal(got, torch.clamp(x, min=0)), "ABY3 ReLU chain: revealed values differ from max(x, 0)"
    # the same chain with the carry tree's messages kept as bit strings between the parties (the *_bits segments the C++
    # engines run; they make the wire bytes on the host): 252 of the 380 B/elem and party are those messages
    be.msb_bits = True
    try:
        aby3_relu()
        dtb, Rb = _timed(torch, aby3_relu)
    finally:
        be.msb_bits = False
    assert all(torch.equal(be.tensor(Rb[j]), be.tensor(R[j])) for j in range(3)), "ABY3 ReLU chain: bit-string messages change the shares"
    rate("aby3_relu_bit_messages_elems_per_s", n, dtb)
    out["aby3_relu_bit_messages_ms"] = round(dtb * 1e3, 3)
    del R, Rb, got

    # ---- MyEngine ReLU chain (test/test_Myengine.cpp, benchmark/benchmark1.cpp:180-204): msb -> bit_inject -> subtract,
    # parties 0 / 1 and the dealer on this GPU; the drivers take whole vectors, so it runs on one rank only
    if world == 1:
        import mpc_amd.my_protocols as mp
        from mpc_amd.my_backends import GpuMy

        class ResidentMy(ResidentBackend, GpuMy):
            pass

        bm = ResidentMy(ctx, torch, dev)
        xm = torch.div(x, 2, rounding_mode="floor")  # msb's domain is [-2^62, 2^62) (test/test_Myengine.cpp: pop / 2)
        xmD = ctx.wrap(xm, np.int64)

        def my_relu():
            w = pr.World(bm, record=False)
            Xs = mp.my_share(w, xmD, n)
            Ab = mp.my_msb(w, Xs)
            C = mp.my_bit_inject(w, Xs, Ab)
            return [bm.ring_sub(Xs[j], C[j]) for j in range(2)]

        my_relu()
        bm.events = []
        dt, R = _timed(torch, my_relu)
        seg = bm.segment_ms()
        bm.events = None
        rate("my_relu_elems_per_s", n, dt)
        out["my_relu_ms"] = round(dt * 1e3, 3)
        out["my_relu_segments_ms"] = {k: round(v, 3) for k, v in sorted(seg.items(), key=lambda kv: -kv[1])}
        got = torch.from_numpy(bm.down(R[0])).to(dev) + torch.from_numpy(bm.down(R[1])).to(dev)
        assert torch.equal(got, torch.clamp(xm, min=0)), "MyEngine ReLU chain: revealed values differ from max(x, 0)"
    ctx.call("set_rand_cache", 4 * GiB)
    out["_counts"] = counts
    return out


def merge_bitwise(per_rank):
    """Whole-job element rates from the ranks' dictionaries: all elements of an item / the slowest rank's time.
    Millisecond entries are rank 0's."""
    errs = [d["error"] for d in per_rank if "error" in d]
    if errs:
        return {"error": "; ".join(errs)}
    out = dict(per_rank[0])
    counts = out.pop("_counts")
    out.pop("n_this_rank", None)
    for key in counts:
        slowest = max(d["_counts"][key] / d[key] for d in per_rank)
        out[key] = sum(d["_counts"][key] for d in per_rank) / slowest
    g = len(per_rank)
    # per-GPU HBM rates on the algorithmic bytes of each item: privateCompare party role 265 B/elem (128 x-bit shares +
    # 8 r + 1 beta in, 128 d out; the s / u bytes come from the generator), helper role 257 B/elem; ABY3 ReLU chain
    # 380 B/elem and party (16 in + 16 out + the bytes the party sends and receives: 182 + 166, 182 + 166, 158 + 190),
    # three parties per element on this GPU
    for label in ("cold", "warm"):
        out[f"pc_party_{label}_GBps_per_gpu"] = 265.0 * out[f"pc_party_{label}_elems_per_s"] / g / 1e9
        out[f"pc_party_{label}_hbm_frac"] = out[f"pc_party_{label}_GBps_per_gpu"] / HBM_PEAK_GBS
    out["pc_helper_GBps_per_gpu"] = 257.0 * out["pc_helper_elems_per_s"] / g / 1e9
    out["pc_helper_hbm_frac"] = out["pc_helper_GBps_per_gpu"] / HBM_PEAK_GBS
    # SecureNN msb: 648 B/elem for each of the roles of parties 0 / 1 (8 in + 8 out + 288 sent + 344 received); the
    # helper's bytes are not counted, so the fraction is a lower bound for the three roles this GPU plays
    for label in ("cold", "warm"):
        out[f"snn_msb_{label}_GBps_per_gpu"] = 2 * 648.0 * out[f"snn_msb_{label}_elems_per_s"] / g / 1e9
        out[f"snn_msb_{label}_hbm_frac"] = out[f"snn_msb_{label}_GBps_per_gpu"] / HBM_PEAK_GBS
    out["aby3_relu_GBps_per_gpu"] = 3 * 380.0 * out["aby3_relu_elems_per_s"] / g / 1e9
    out["aby3_relu_hbm_frac"] = out["aby3_relu_GBps_per_gpu"] / HBM_PEAK_GBS
    out["ranks"] = g
    # HBM bytes per launch from the committed rocprofv3 --pmc passes (taken at 2^24 elements on one GPU)
    path = os.path.join(ROOT, PMC_BITWISE)
    if g == 1 and out.get("n") == 1 << 24 and os.path.exists(path):
        pmc = json.load(open(path))
        for k in ("pc_party_kernel", "pc_helper_kernel"):
            out[k.replace("_kernel", "") + "_traffic"] = {
                "measured": pmc[k]["hbm_read_bytes_per_launch"] + pmc[k]["hbm_write_bytes_per_launch"],
                "algorithmic": pmc[k]["algorithmic_bytes_per_launch"], "unit": "bytes per launch"}
    return out



# ---------------------------------------------------------------------------------------------------------------
# The other configurations of the metric, measured after the headline outside its timed region (rank 0, one GPU)
MLP_DIMS = [(128, 784, 128), (128, 128, 128), (128, 128, 10)]


def _mlp_plain(X, Ws):
    """The plaintext 784-128-128-10 network with the fixed-point rule of the share-local rescale."""
    h = X.reshape(128, 784)
    for li, (_, l, c) in enumerate(MLP_DIMS):
        h = (h @ Ws[li].reshape(l, c)) >> 16
        if li < 2:
            h = np.maximum(h, 0)
    return h


def _mlp_forward(pr, be, w, act, Wsh, op):
    """Per layer matrix_mul -> local >> 16 on each share -> msb -> vector_mul(a, msb) -> a - that, no ReLU after the
    last layer (benchmark/benchmark1.cpp:104-136); op(layer, name, fn) runs and times one operator."""
    for li, (r, l, c) in enumerate(MLP_DIMS):
        Z = op(li, "matrix_mul", lambda: pr.snn_matmul(w, act, Wsh[li], r, l, c))
        a = op(li, "shift16", lambda: [be.ring_sar(Z[0], 16), be.ring_sar(Z[1], 16)])
        if li < 2:
            m = op(li, "msb", lambda: pr.snn_msb(w, a))
            am = op(li, "vector_mul", lambda: pr.snn_vecmul(w, a, m))
            a = op(li, "subtract", lambda: [be.ring_sub(a[j], am[j]) for j in range(2)])
        act = a
    return act


def config4_mlp(ctx, torch, dev, reps=5):
    """BASELINE configs[3]: the SecureNN 784-128-128-10 MLP at batch 128 (input and W1..W3 = PRgenerator(24,0)
    .pop_int64() >> 48 in that order), all three roles on this GPU, every share and message resident: end-to-end latency
    of the three layers (wall clock between two device synchronisations; the input and the weights are shared before)
    and the device time per layer and operator."""
    import copy

    import mpc_amd.protocols as pr
    from mpc_amd.backends import ResidentBackend
    be = ResidentBackend(ctx, torch, dev)
    total = 128 * 784 + sum(l * c for _, l, c in MLP_DIMS)
    TableBuilder(ctx, None, 0, 1).build([(24, total, 0), (KEYS, 0, 5 * 128 * 784 * 128)])
    words = torch.empty(total, dtype=torch.int64, device=dev)
    ctx.call("prg_fill_pop", 24, 0, 0, total, words)
    words >>= 48
    X, off, Ws = words[:128 * 784].clone(), 128 * 784, []
    for _, l, c in MLP_DIMS:
        Ws.append(words[off:off + l * c].clone())
        off += l * c
    w = pr.World(be, record=False)
    act0 = pr.snn_share(w, ctx.wrap(X, np.int64), X.numel())[:2]
    Wsh = [pr.snn_share(w, ctx.wrap(Ws[li], np.int64), Ws[li].numel())[:2] for li in range(3)]
    pops0 = copy.deepcopy(w.pops)
    plain = lambda layer, name, fn: fn()
    ctx.call("set_rand_cache", 4 << 30)
    for _ in range(2):  # warm-up: tables, byte cache, allocator
        w.pops = copy.deepcopy(pops0)
        out = _mlp_forward(pr, be, w, act0, Wsh, plain)
    lat = []
    for _ in range(reps):
        w.pops = copy.deepcopy(pops0)
        dt, out = _timed(torch, lambda: _mlp_forward(pr, be, w, act0, Wsh, plain))
        lat.append(dt * 1e3)
    ops = []

    def timed_op(layer, name, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        ops.append((layer, name, e0, e1))
        return r

    w.pops = copy.deepcopy(pops0)
    out = _mlp_forward(pr, be, w, act0, Wsh, timed_op)
    torch.cuda.synchronize()
    per_op = {}
    for layer, name, e0, e1 in ops:
        per_op.setdefault(f"layer{layer + 1}", {})[name] = round(e0.elapsed_time(e1), 4)
    got = (be.down(out[0]) + be.down(out[1])).reshape(128, 10)
    want = _mlp_plain(X.cpu().numpy(), [t.cpu().numpy() for t in Ws])
    assert np.abs(got - want).max() <= 64, "config 4: the revealed output differs from the plaintext network"
    lat.sort()
    # The same forward pass captured once into a HIP graph and replayed: the ~150 small kernels of the three layers go
    # to the GPU as one submission (after the warm-up passes the library allocates nothing and synchronises nothing, so
    # its calls can be captured as they are).  The replay must give the eager pass's shares bit for bit.
    graph = {}
    try:
        main_stream = torch.cuda.current_stream()
        side = torch.cuda.Stream(dev)
        side.wait_stream(main_stream)
        with torch.cuda.stream(side):
            ctx.call("set_stream", side.cuda_stream)
            for _ in range(2):  # this stream's blocks in torch's allocator
                w.pops = copy.deepcopy(pops0)
                _mlp_forward(pr, be, w, act0, Wsh, plain)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            w.pops = copy.deepcopy(pops0)
            gout = _mlp_forward(pr, be, w, act0, Wsh, plain)
        torch.cuda.synchronize()
        glat = []
        for _ in range(reps + 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g.replay()
            torch.cuda.synchronize()
            glat.append((time.perf_counter() - t0) * 1e3)
        glat = sorted(glat[2:])
        same = all(np.array_equal(be.down(gout[j]), be.down(out[j])) for j in range(2))
        assert same, "config 4: the graph replay's shares differ from the eager pass"
        graph = {"graph_latency_ms": round(glat[len(glat) // 2], 3), "graph_latency_ms_min": round(glat[0], 3),
                 "graph_verified": "replayed shares equal the eager pass bit for bit"}
        del g, gout
    except AssertionError:
        raise
    except Exception as ex:  # capture is an extra: the eager numbers stand without it
        graph = {"graph_error": f"{type(ex).__name__}: {str(ex)[:200]}"}
    finally:
        ctx.call("set_stream", torch.cuda.current_stream().cuda_stream)
    res = {"workload": "SecureNN MLP 784-128-128-10, batch 128, Q16 inputs from PRgenerator(24,0); per layer matrix_mul, "
                        ">>16, msb, vector_mul, subtract; 3 roles on one GPU, shares and messages resident",
            "latency_ms": round(lat[len(lat) // 2], 3), "latency_ms_min": round(lat[0], 3), "reps": reps,
            "device_ms_per_op": per_op,
            "device_ms_by_op": {k: round(sum(v.get(k, 0.0) for v in per_op.values()), 4)
                                for k in ("matrix_mul", "shift16", "msb", "vector_mul", "subtract")},
            "verified": "revealed output within 64 units of the plaintext network (one unit per share-local shift)"}
    res.update(graph)
    return res


def cpu_mlp():
    """The CPU oracle on configuration 4: one thread plays the three roles in turn, as config4_mlp does on the GPU."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))  # the oracle backend is test infrastructure: cpu_baseline leg only
    import mpc_amd.protocols as pr
    from backends import OracleBackend
    orc = OracleBackend()
    total = 128 * 784 + sum(l * c for _, l, c in MLP_DIMS)
    words = orc.prg_fill_pop(24, 0, total) >> 48
    X, off, Ws = words[:128 * 784].copy(), 128 * 784, []
    for _, l, c in MLP_DIMS:
        Ws.append(words[off:off + l * c].copy())
        off += l * c
    w = pr.World(orc, record=False)
    act0 = pr.snn_share(w, X, X.size)[:2]
    Wsh = [pr.snn_share(w, Ws[li], Ws[li].size)[:2] for li in range(3)]
    t0 = time.perf_counter()
    out = _mlp_forward(pr, orc, w, act0, Wsh, lambda layer, name, fn: fn())
    dt = time.perf_counter() - t0
    assert np.abs((out[0] + out[1]).reshape(128, 10) - _mlp_plain(X, Ws)).max() <= 64
    return {"value": round(dt * 1e3, 2), "unit": "ms latency", "cores": 1, "kind": "port",
            "sample": "the whole configuration (batch 128), three roles in turn on one thread"}


def config5_slice(ctx, torch, dev):
    """BASELINE configs[4] on one GPU: (a) the row block one of eight GPUs owns of the 16384 x 16384 ABY3 share matmul
    (rows [0, 2048) of X, Y replicated: 2048 x 16384 . 16384 x 16384, four accumulation chunks per product; X and Y from
    PRgenerator(20,0), shared by share_matrix on the device; party 0's matrix_mul segment); (b) the ReLU chain
    add_prepare -> msb -> bit_inject -> subtract on 2^28 elements from PRgenerator(21,0), as sixteen element ranges of
    2^24 (n_total = 2^28, i0 = 2^24 b): the documented batch schedule; all three parties on this GPU."""
    import mpc_amd.protocols as pr
    from mpc_amd.backends import ResidentBackend
    full, rows, row0 = 16384, 2048, 0
    l = c = full
    plan = aby3_matmul_pop_plan(full, l, c)
    nt, n = 1 << 28, 1 << 24
    t0 = time.perf_counter()
    TableBuilder(ctx, None, 0, 1).build([(KEY01, max(plan["matmul"] + rows * c, 7 * nt), 0), (KEY20, max(plan["matmul"] + rows * c, 7 * nt), 0),
                                         (KEY12, max(plan["matmul"], 7 * nt), 0), (20, full * l + l * c, 0), (21, nt, 0)])
    out = {"ckpt_build_s": round(time.perf_counter() - t0, 2)}
    e = lambda k: torch.empty(k, dtype=torch.int64, device=dev)
    X, Y = e(rows * l), e(l * c)
    ctx.call("prg_fill_pop", 20, 0, row0 * l, rows * l, X)
    ctx.call("prg_fill_pop", 20, 0, full * l, l * c, Y)
    RX, RY = [e(rows * l), e(rows * l)], [e(l * c), e(l * c)]
    for j in range(2):  # parties 0 and 1: party 0's product reads both
        sn, sp = pair_seed(j, (j + 1) % 3), pair_seed(j, (j + 2) % 3)
        ctx.call("aby3_share_local", X if j == 0 else None, rows * l, sn, row0 * l, sp, row0 * l, RX[j])
        ctx.call("aby3_share_local", Y if j == 0 else None, l * c, sn, plan["share_y"], sp, plan["share_y"], RY[j])
    del X, Y
    Z = e(rows * c)
    call = lambda: ctx.call("aby3_matmul_local_rows", RX[0], RX[1], RY[0], RY[1], rows, l, c, row0, KEY01, plan["matmul"],
                            KEY20, plan["matmul"], Z)
    call()
    torch.cuda.synchronize()
    reps = 3
    ctx.call("profile", 1)
    dt, _ = _timed(torch, call, reps=reps)
    gemm_ms, gemm_cnt = prof_read(ctx, "limb_gemm")
    split_ms, _ = prof_read(ctx, "split_limbs")
    ctx.call("profile", 0)
    import mpc_amd
    lib = mpc_amd.load_library()
    dims = (ctypes.c_uint64(rows), ctypes.c_uint64(2 * l), ctypes.c_uint64(c))
    nprod = lib.mpcx_ring_gemm_products(*dims)
    lib.mpcx_ring_gemm_kernel_name.restype = ctypes.c_char_p
    ops = 36 * 2 * rows * c * (2 * l)  # algorithmic int8 operations of the block's product
    alg = ops * reps / (gemm_ms * 1e-3) / 1e12
    # the product without its mask passes a Freivalds check: P v = X0 ((Y0 + Y1) v) + X1 (Y0 v) over Z_2^64
    ma, mb = e(rows * c), e(rows * c)
    ctx.call("prg_fill_pop", KEY01, 0, plan["matmul"] + row0 * c, rows * c, ma)
    ctx.call("prg_fill_pop", KEY20, 0, plan["matmul"] + row0 * c, rows * c, mb)
    torch.cuda.synchronize()
    P = (Z - (ma - mb)).view(rows, c)
    del ma, mb
    matvec = lambda M, v: (M * v.unsqueeze(0)).sum(1)  # int64 arithmetic on the device wraps mod 2^64
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    v = torch.randint(-2 ** 63, 2 ** 63 - 1, (c,), dtype=torch.int64, device=dev, generator=g) | 1
    ysv = matvec(RY[0].view(l, c), v) + matvec(RY[1].view(l, c), v)
    y0v = matvec(RY[0].view(l, c), v)
    assert torch.equal(matvec(P, v), matvec(RX[0].view(rows, l), ysv) + matvec(RX[1].view(rows, l), y0v)), \
        "config 5 row block: P v != X0 ((Y0 + Y1) v) + X1 (Y0 v)"
    # the same call with Y as a resident second operand (weights that many products use): its residue planes are made
    # once and the product splits only X
    op = ctypes.c_void_p()
    U64 = ctypes.c_uint64
    ctx._call("mpcx_aby3_operand_prepare", ctypes.c_void_p(RY[0].data_ptr()), ctypes.c_void_p(RY[1].data_ptr()), U64(l), U64(c),
              ctypes.c_int(0), ctypes.byref(op))
    Zp = e(rows * c)
    prepared = lambda: ctx._call("mpcx_aby3_matmul_local_prepared", ctypes.c_void_p(RX[0].data_ptr()),
                                 ctypes.c_void_p(RX[1].data_ptr()), ctypes.c_int(0), op, U64(rows), U64(row0), U64(KEY01),
                                 U64(plan["matmul"]), U64(KEY20), U64(plan["matmul"]), ctypes.c_void_p(Zp.data_ptr()))
    prepared()
    dtp, _ = _timed(torch, prepared, reps=reps)
    assert torch.equal(Zp, Z), "config 5 row block: the product with a prepared operand differs"
    lib.mpcx_aby3_operand_bytes.restype = ctypes.c_uint64
    plane_bytes = lib.mpcx_aby3_operand_bytes(op)
    ctx._call("mpcx_aby3_operand_free", op)
    del P, RX, RY, Z, Zp, ysv, y0v
    torch.cuda.empty_cache()
    out["matmul_row_block"] = {
        "shape": f"{rows}x{l} . {l}x{c} (rows [{row0}, {row0 + rows}) of {full}), one party's matrix_mul segment",
        "ms": round(dt * 1e3, 3), "gemm_ms": round(gemm_ms / reps, 3), "split_combine_ms": round(split_ms / reps, 3),
        "gemm_launches_per_call": gemm_cnt // reps, "kernel": lib.mpcx_ring_gemm_kernel_name(*dims).decode(),
        "algorithmic_TOPs": round(alg, 1), "frac": round(alg / INT8_PEAK_TOPS, 4),
        "issued_frac": round(alg * nprod / 36 / INT8_PEAK_TOPS, 4),
        "call_frac": round(ops / dt / 1e12 / INT8_PEAK_TOPS, 4),
        "ring_Gops": round(3 * 2 * rows * l * c / dt / 1e9, 1), "verified": "Freivalds check of the unmasked block",
        "ms_second_operand_prepared": round(dtp * 1e3, 3), "prepared_call_frac": round(ops / dtp / 1e12 / INT8_PEAK_TOPS, 4),
        "prepared_planes_GiB": round(plane_bytes / 2 ** 30, 2)}

    be = ResidentBackend(ctx, torch, dev)
    x = torch.empty(n, dtype=torch.int64, device=dev)
    T = be.tensor
    spent, per_batch = 0.0, []
    for b in range(nt // n):
        i0 = b * n
        ctx.call("prg_fill_pop", 21, 0, i0, n, x)
        xD = ctx.wrap(x, np.int64)

        def chain():
            w = pr.World(be, record=False, n_total=nt, i0=i0)
            Xs = pr.aby3_share(w, xD, n)
            Gs, Ps = pr.aby3_add_prepare(w, Xs)
            Zb = pr.aby3_msb(w, Gs, Ps)
            return pr.aby3_bit_inject(w, Xs, Zb, minuend=Xs)

        if b == 0:
            chain()  # warm-up of the allocator
        dt, R = _timed(torch, chain)
        spent += dt
        per_batch.append(round(dt * 1e3, 3))
        relu = T(R[0][0]) + T(R[1][0]) + T(R[2][0])
        assert torch.equal(relu, torch.clamp(x, min=0)), f"config 5 ReLU chain: range {b} does not reveal max(x, 0)"
        del R, relu
    rate = nt / spent
    out["relu_2_28"] = {"n": nt, "schedule": "16 element ranges of 2^24 (n_total = 2^28, i0 = 2^24 b), 3 parties on one GPU",
                        "elems_per_s": rate, "ms_total": round(spent * 1e3, 2), "ms_per_range": per_batch,
                        "GBps": round(3 * 380.0 * rate / 1e9, 1), "hbm_frac": round(3 * 380.0 * rate / 1e9 / HBM_PEAK_GBS, 4),
                        "verified": "every range reveals max(x, 0)"}
    # the same schedule with the carry tree's messages kept as bit strings between the parties (what the C++ engines'
    # device side does; the wire bytes are then made on the host)
    be.msb_bits = True
    try:
        spent_b = 0.0
        for b in range(nt // n):
            i0 = b * n
            ctx.call("prg_fill_pop", 21, 0, i0, n, x)
            xD = ctx.wrap(x, np.int64)

            def chain_b():
                w = pr.World(be, record=False, n_total=nt, i0=i0)
                Xs = pr.aby3_share(w, xD, n)
                Gs, Ps = pr.aby3_add_prepare(w, Xs)
                Zb = pr.aby3_msb(w, Gs, Ps)
                return pr.aby3_bit_inject(w, Xs, Zb, minuend=Xs)

            if b == 0:
                chain_b()
            dt, R = _timed(torch, chain_b)
            spent_b += dt
            assert torch.equal(T(R[0][0]) + T(R[1][0]) + T(R[2][0]), torch.clamp(x, min=0)), \
                f"config 5 ReLU chain with bit-string messages: range {b} does not reveal max(x, 0)"
            del R
    finally:
        be.msb_bits = False
    out["relu_2_28"]["bit_messages_elems_per_s"] = nt / spent_b
    out["relu_2_28"]["bit_messages_ms_total"] = round(spent_b * 1e3, 2)
    return out


def config5_full(ctx, torch, dev, n=16384, steps=2):
    """BASELINE configs[4]'s matrix product whole on this GPU (it fits: ~30 GiB of shares for the three parties): the
    headline's step -- share_matrix x2, matrix_mul, reveal_matrix, three parties -- at 16384 x 16384, every product as
    four accumulation chunks of depth 8192; verified like the headline (equal reveals, sampled exact entries,
    Z v = X (Y v) over Z_2^64)."""
    job = Aby3MatmulJob(ctx, torch, dev, n, 0, 1, gather=False)
    job.step()
    torch.cuda.synchronize()
    ctx._call("mpcx_profile_select", b"limb_gemm")
    ctx.call("profile", 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        job.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    gemm_ms, gemm_cnt = prof_read(ctx, "limb_gemm")
    ctx.call("profile", 0)
    ctx._call("mpcx_profile_select", None)
    job.verify()
    ops = 3 * 36 * 2 * n * n * (2 * n)  # algorithmic int8 operations of a step (three products of depth 2n)
    out = {"workload": f"ABY3 {n}x{n} . {n}x{n} share matmul + reconstruct, 3 parties per step, one GPU, "
                       f"{2 * n // 8192} accumulation chunks per product",
           "ms_per_step": round(dt * 1e3, 2), "steps": steps, "ring_Gops": round(job.ring_ops_per_step() / dt / 1e9, 1),
           "gemm_ms_per_step": round(gemm_ms / steps, 2), "gemm_launches_per_step": gemm_cnt // steps,
           "frac": round(ops * steps / (gemm_ms * 1e-3) / 1e12 / INT8_PEAK_TOPS, 4),
           "step_frac": round(ops / dt / 1e12 / INT8_PEAK_TOPS, 4),
           "verified": "equal reveals, 24 exact entries, Z v = X (Y v)"}
    del job
    torch.cuda.empty_cache()
    return out


def prg_rates(ctx, torch, dev, log2n=28):
    """The generator kernel by itself: mpcx_prg_fill_pop of 2^28 words (8 bytes written per word), and the same
    regeneration fused into a consumer (the ABY3 share mask pop(next) - pop(prev): two stream words per 8 bytes
    written).  CPU beside it: the reference's own PRgenerator (oracle/_ref/libref_prg.so, built from
    utils/PRgenerator.cpp) popping 2^27 words on one host core."""
    n = 1 << log2n
    t0 = time.perf_counter()
    TableBuilder(ctx, None, 0, 1).build([(KEY01, n, 0), (KEY20, n, 0)])
    out = {"words": n, "ckpt_build_s": round(time.perf_counter() - t0, 2)}
    buf = torch.empty(n, dtype=torch.int64, device=dev)
    ctx.call("prg_fill_pop", KEY01, 0, 0, n, buf)
    dt, _ = _timed(torch, lambda: ctx.call("prg_fill_pop", KEY01, 0, 0, n, buf), reps=3)
    out["fill_words_per_s"] = n / dt
    out["fill_GBps"] = round(8 * n / dt / 1e9, 1)
    out["fill_hbm_frac"] = round(8 * n / dt / 1e9 / HBM_PEAK_GBS, 4)
    ctx.call("aby3_share_local", None, n, KEY01, 0, KEY20, 0, buf)
    dt, _ = _timed(torch, lambda: ctx.call("aby3_share_local", None, n, KEY01, 0, KEY20, 0, buf), reps=3)
    out["fused_words_per_s"] = 2 * n / dt
    out["fused_GBps_written"] = round(8 * n / dt / 1e9, 1)
    # known answer: the first pops of the KEY01 stream (utils/PRgenerator.cpp:34-37)
    ctx.call("prg_fill_pop", KEY01, 0, 0, n, buf)
    first = [int(v) & (2 ** 64 - 1) for v in buf[:3].cpu().tolist()]
    assert first == [0x2643CB33EC326FF6, 0x98BEFDA8E0849415, 0x64A5D429C2A35948], "PRG known-answer words differ"
    return out


def cpu_prg(log2n=27):
    ref = os.path.join(ROOT, "oracle", "_ref", "libref_prg.so")
    if not os.path.exists(ref):
        return None
    lib = ctypes.CDLL(ref)
    n = 1 << log2n
    out = np.empty(n, dtype=np.int64)
    t0 = time.perf_counter()
    lib.ref_prg_pop(ctypes.c_uint64(KEY01), ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(n),
                    ctypes.c_void_p(out.ctypes.data))
    dt = time.perf_counter() - t0
    assert int(out[0]) & (2 ** 64 - 1) == 0x2643CB33EC326FF6
    return {"value": n / dt, "unit": "words/s", "cores": 1, "kind": "reference",
            "sample": f"PRgenerator(0xAB01, 0) constructed and popped 2^{log2n} times (utils/PRgenerator.cpp), {dt:.2f} s"}


def ckpt_cache_times():
    """Wall time of building the checkpoint table of a 2^28-word stream in a fresh process: walked (no cache), and read
    from the on-disk cache a previous process left."""
    import subprocess
    import tempfile
    prog = ("import sys, time; sys.path.insert(0, %r); from mpc_amd import Context; c = Context(0); t0 = time.perf_counter(); "
            "c.call('prg_ckpt_build', 0xAB01, 0, 1 << 28, 0); print('BUILD_S', time.perf_counter() - t0)" % ROOT)
    out = {}
    with tempfile.TemporaryDirectory(prefix="mpcx_ckpt_") as d:
        for label in ("cold", "cached"):
            r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300,
                               env=dict(os.environ, MPCX_CKPT_CACHE=d))
            got = [l for l in r.stdout.splitlines() if l.startswith("BUILD_S")]
            if r.returncode or not got:
                return {"error": (r.stderr or r.stdout)[-300:]}
            out[label + "_s"] = round(float(got[0].split()[1]), 3)
    out["what"] = "mpcx_prg_ckpt_build of a 2^28-word pop stream in a fresh process, MPCX_CKPT_CACHE empty / filled"
    return out


def host_engine_rates():
    """The PCIe- and host-inclusive rates: the C++ drop-in engines (mpc_amd/host/benchmark1: three party threads over
    InProcMessenger, std::string messages, host containers in and out of every method, one GPU shared by the parties)."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "mpc_amd", "host", "benchmark1")
    if not os.path.exists(exe):
        return {"error": "mpc_amd/host/benchmark1 is not built"}
    runs = [("aby3_4096", ["--engine", "aby3", "--width", "4096", "--vec", "1048576", "--repeat", "3", "--no-mlp"]),
            ("aby3_4096_second_operand_pinned", ["--engine", "aby3", "--width", "4096", "--vec", "4096", "--repeat", "3", "--no-mlp", "--pin"]),
            ("securenn_1024", ["--engine", "securenn", "--width", "1024", "--vec", "1048576", "--repeat", "3", "--no-mlp"]),
            ("mlp", ["--engine", "securenn", "--mlp-only", "--repeat", "3"])]
    out = {}
    for tag, args in runs:
        t0 = time.perf_counter()
        try:
            r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=240)
        except subprocess.TimeoutExpired:
            out[tag] = {"error": "timeout"}
            continue
        if r.returncode != 0 or "all results verified" not in r.stdout:
            out[tag] = {"error": (r.stdout + r.stderr)[-300:]}
            continue
        rec = {"wall_s": round(time.perf_counter() - t0, 1), "args": " ".join(args), "verified": True}
        for line in r.stdout.splitlines():
            if not line.startswith("RESULT\t"):
                continue
            f = line.split("\t")
            key = re.sub(r"[^a-z0-9]+", "_", f[1].lower()).strip("_")
            kv = dict(x.split("=", 1) for x in f[2:])
            rec[key] = {"ms": float(kv["ms"])}
            if "rate" in kv:
                rec[key]["rate"] = float(kv["rate"])
                rec[key]["unit"] = kv["unit"]
            if "layers_ms" in kv:
                rec[key]["layers_ms"] = [float(x) for x in kv["layers_ms"].split(",")]
        out[tag] = rec
    return out


def config5_multi(ctx, torch, dev, dist, rank, world, tables, gather_to, host_gather, full=16384, log2nt=28):
    """BASELINE configs[4] over the ranks of a node (strong scaling: the work is fixed, every rank takes 1 / world of it).
    (a) ABY3 16384 x 16384 share matmul: rank g owns rows [16384 g / world, ...) of X, Y is replicated, every product
    runs as four accumulation chunks, and the row blocks of each party's message are gathered to rank 0 over RCCL
    exactly as in the headline job (verified the same way).  (b) The 2^28-element ReLU chain in the sixteen-range
    schedule of 2^24 elements each: rank g runs ranges [16 g / world, 16 (g + 1) / world), all three parties on its GPU."""
    import mpc_amd.protocols as pr
    from mpc_amd.backends import ResidentBackend
    out = {"scaling": "strong", "ranks": world}
    job = Aby3MatmulJob(ctx, torch, dev, full, rank, world, host_gather=host_gather, rows=full // world, gather=True,
                        gather_to=gather_to, tables=tables)

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    job.step()
    job.drain()
    barrier()
    steps = 2
    t0 = time.perf_counter()
    for _ in range(steps):
        job.step()
    job.drain()
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev if not host_gather else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    job.verify()
    ops_alg = 3 * 36 * 2 * full * full * (2 * full)  # three parties, one product of depth 2 l each
    out["matmul"] = {"workload": f"ABY3 {full}x{full} . {full}x{full} share matmul + reconstruct, 3 parties per step, "
                                 f"{full // world} rows per rank, Y replicated, {max(1, 2 * full // 8192)} accumulation chunk(s) "
                                 "per product, gather to "
                                 + ("rank 0" if gather_to == "root" else "every rank"),
                     "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps,
                     "ring_Gops": round(3 * 3 * 2 * full ** 3 / (dt / steps) / 1e9, 1),
                     "algorithmic_TOPs_whole_job": round(ops_alg / (dt / steps) / 1e12, 1),
                     "step_frac_of_peak": round(ops_alg / (dt / steps) / 1e12 / (INT8_PEAK_TOPS * world), 4)}
    del job
    torch.cuda.empty_cache()

    be = ResidentBackend(ctx, torch, dev)
    nt, nb = 1 << log2nt, 16
    n = nt // nb
    b0, cnt = split_range(nb, world, rank)
    x = torch.empty(n, dtype=torch.int64, device=dev)
    T = be.tensor
    spent, ok = 0.0, True
    for b in range(b0, b0 + cnt):
        i0 = b * n
        ctx.call("prg_fill_pop", 21, 0, i0, n, x)
        xD = ctx.wrap(x, np.int64)

        def chain():
            w = pr.World(be, record=False, n_total=nt, i0=i0)
            Xs = pr.aby3_share(w, xD, n)
            Gs, Ps = pr.aby3_add_prepare(w, Xs)
            Zb = pr.aby3_msb(w, Gs, Ps)
            return pr.aby3_bit_inject(w, Xs, Zb, minuend=Xs)

        if b == b0:
            chain()
        d, R = _timed(torch, chain)
        spent += d
        ok = ok and bool(torch.equal(T(R[0][0]) + T(R[1][0]) + T(R[2][0]), torch.clamp(x, min=0)))
        del R
    per_rank = [None] * world
    dist.all_gather_object(per_rank, (spent, cnt * n, ok))
    assert all(p[2] for p in per_rank), "config 5 ReLU chain: a range does not reveal max(x, 0)"
    slowest = max(p[0] for p in per_rank)
    rate = nt / slowest
    out["relu_2_28"] = {"n": nt, "schedule": f"16 element ranges of 2^{log2nt - 4}, {nb / world:g} per rank",
                        "elems_per_s": rate, "ms_slowest_rank": round(slowest * 1e3, 2),
                        "hbm_frac_per_gpu": round(3 * 380.0 * rate / world / 1e9 / HBM_PEAK_GBS, 4),
                        "verified": "every range reveals max(x, 0)"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=4096, help="matrix dimension per GPU (rows of the row-block, l and c)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-bitwise", action="store_true")
    ap.add_argument("--config5-size", type=int, default=16384, help="rehearsals: a smaller matrix for the multi-rank "
                    "record of configuration 5")
    ap.add_argument("--config5-log2n", type=int, default=28, help="rehearsals: a shorter vector for its ReLU chain")
    ap.add_argument("--no-extras", action="store_true", help="skip the records of configurations 4 and 5, the generator "
                    "rates and the host-engine rates")
    ap.add_argument("--no-verify", action="store_true", help="development only: skip the result check")
    ap.add_argument("--bitwise-log2n", type=int, default=24, help="elements of the privateCompare / msb / ReLU runs")
    ap.add_argument("--rows", type=int, default=0, help="rows of X per GPU when different from --size (a row block "
                    "of a larger product, e.g. --size 16384 --rows 2048 = one GPU's share of the 16384^2 configuration)")
    ap.add_argument("--gather", default="root", choices=["root", "all"],
                    help="who receives the row blocks of a party's message: the messenger-facing rank 0, or every rank")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the multi-rank path on fewer GPUs than ranks (gathers through the host)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dev_index = local_rank if args.dist_backend == "nccl" else local_rank % torch.cuda.device_count()
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1 or "RANK" in os.environ:  # under torch.distributed.run the collectives run even with one rank
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    if dist is not None and args.dist_backend == "nccl":
        # bring the communicator up outside the timed region whatever --warmup is
        probe = torch.zeros(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(probe, probe[rank:rank + 1].clone())
        torch.cuda.synchronize()

    from mpc_amd import Context
    ctx = Context(dev_index)
    ctx.call("set_stream", torch.cuda.current_stream().cuda_stream)

    tables = TableBuilder(ctx, dist, rank, world)
    job = Aby3MatmulJob(ctx, torch, dev, args.size, rank, world, host_gather=args.dist_backend == "gloo", rows=args.rows or None,
                        gather=dist is not None, gather_to=args.gather, tables=tables)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Every HIP event the library records costs the stream ~10 us between two kernels.  The per-family breakdown is
    # therefore taken on the last warm-up step; in the timed region only the dominant kernel (the GEMM) is bracketed.
    families = ("limb_gemm", "split_limbs", "prg_fill")
    breakdown = None
    for w in range(args.warmup):
        if w == args.warmup - 1:  # the last warm-up step (the first one also pays for allocations and table uploads)
            ctx.call("profile", 1)
        job.step()
    job.drain()
    barrier()
    if args.warmup:
        breakdown = {k: round(prof_read(ctx, k)[0], 3) for k in families}
        ctx._call("mpcx_profile_select", b"limb_gemm")
    ctx.call("profile", 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.step()
    job.drain()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    gemm_ms, gemm_cnt = prof_read(ctx, "limb_gemm")
    if breakdown is None:  # no warm-up: all families were recorded in the timed region
        breakdown = {k: round(prof_read(ctx, k)[0] / args.steps, 3) for k in families}
    breakdown["limb_gemm"] = round(gemm_ms / args.steps, 3)  # the timed region's own launches
    ctx.call("profile", 0)
    ctx._call("mpcx_profile_select", None)
    if not args.no_verify:
        job.verify()

    bitwise = None
    if not args.no_bitwise:
        job = job.shape_only()  # frees the matmul operands before the element-wise runs allocate theirs
        torch.cuda.empty_cache()
        if tables.shared:
            tables.build(bitwise_table_specs(args.bitwise_log2n, world))
        try:
            mine = bitwise_rates(ctx, torch, dev, args.bitwise_log2n, rank, world, tables)
        except Exception as ex:  # secondary metric must not hide the headline number
            mine = {"error": f"rank {rank}: {ex}"}
        bitwise = [mine]
        if dist is not None:  # no collective inside bitwise_rates: a rank that failed still arrives here
            bitwise = [None] * world
            dist.all_gather_object(bitwise, mine)
    extras = {}
    if not args.no_extras:
        job = job.shape_only()
        torch.cuda.empty_cache()
        if world > 1:
            c5, lg = args.config5_size, args.config5_log2n
            tables.build([(k, max(3 * c5 * c5, 7 << lg), 0) for k in (KEY01, KEY12, KEY20)] + [(20, 2 * c5 * c5, 0), (21, 1 << lg, 0)])
            extras["config5"] = config5_multi(ctx, torch, dev, dist, rank, world, tables, args.gather,
                                              args.dist_backend == "gloo", full=c5, log2nt=lg)
        else:
            for key, fn in (("config4", lambda: config4_mlp(ctx, torch, dev)), ("prg", lambda: prg_rates(ctx, torch, dev)),
                            ("config5_slice", lambda: config5_slice(ctx, torch, dev)),
                            ("config5_full", lambda: config5_full(ctx, torch, dev))):
                try:
                    extras[key] = fn()
                except Exception as ex:  # a secondary record must not hide the headline number
                    extras[key] = {"error": f"{type(ex).__name__}: {ex}"}
                torch.cuda.empty_cache()
            ctx.call("set_rand_cache", 0)  # the host engines below share this GPU
            ctx.call("set_rand_cache", 4 << 30)
            if not args.no_cpu:
                extras["host_engine"] = host_engine_rates()
                extras["ckpt_cache"] = ckpt_cache_times()
    if rank == 0:
        value = job.ring_ops_per_step() * args.steps / dt / 1e9
        # dominant kernel: the int8 MFMA GEMM.  Algorithmic work per product (the positional formulation the metric is
        # defined on): 36 digit pairs x 2 ops x rows x c x K with K = 2l (the two ring products X0*(Y0+Y1) and X1*Y0
        # concatenated along K).  The library may reach the same exact result with fewer int8 products (18 residue
        # products + CRT): `issued` is what the matrix cores actually executed.
        # (products with 2l > 8192 run as several K-chunk launches: count the work per product, not per launch)
        ops_per_product = 36 * 2 * job.rows * job.c * (2 * job.l)
        import mpc_amd
        # matrix_mul is ONE product of depth 2l (X0 (Y0 + Y1) and X1 Y0 concatenated along K): ask for that shape
        lib = mpc_amd.load_library()
        dims = (ctypes.c_uint64(job.rows), ctypes.c_uint64(2 * job.l), ctypes.c_uint64(job.c))
        nprod = lib.mpcx_ring_gemm_products(*dims)
        lib.mpcx_ring_gemm_kernel_name.restype = ctypes.c_char_p
        kernel = lib.mpcx_ring_gemm_kernel_name(*dims).decode()
        avg_s = gemm_ms / gemm_cnt * 1e-3
        achieved = ops_per_product * 3 * args.steps / (gemm_ms * 1e-3) / 1e12
        issued = achieved * nprod / 36
        # the whole product phase: operand split + GEMM + recombination (the residue formulation pays for its 18
        # instead of 36 products with a residue split and a CRT pass; the digit-pair formulation folds in its epilogue)
        phase_ms = gemm_ms + breakdown["split_limbs"] * args.steps
        phase = ops_per_product * 3 * args.steps / (phase_ms * 1e-3) / 1e12
        res = {
            "metric": "Z_2^64 share-matmul Gop/s + MSB/privateCompare elems/sec at 1/2/4/8 GPUs",
            "value": value, "unit": "Gop/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 ring via int8 residues or digits (int32 accumulate)", "data": "synthetic",
            "config": {"workload": f"ABY3 {job.rows_total}x{job.l} . {job.l}x{job.c} Z_2^64 share matmul + reconstruct, "
                                   "3 parties per step (share_matrix x2, matrix_mul local, reveal_matrix), "
                                   "inputs from PRgenerator(20,0)",
                       "rows_per_gpu": job.rows, "parallelism": f"row-block x{world}" if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": INT8_PEAK_TOPS,
                         "unit": "TOP/s", "frac": achieved / INT8_PEAK_TOPS,
                         "step_frac": ops_per_product * 3 * args.steps / dt / 1e12 / (INT8_PEAK_TOPS * world),
                         "issued": issued, "issued_frac": issued / INT8_PEAK_TOPS, "int8_products_per_multiply_add": nprod,
                         "gemm_phase": phase, "gemm_phase_frac": phase / INT8_PEAK_TOPS,
                         "traffic": measured_traffic(args.size, world, kernel),
                         "traffic_unit": f"bytes per launch ({PMC_SUMMARY}: FETCH_SIZE x 2 + WRITE_SIZE, counted between "
                                         "the L2s and the fabric, i.e. in front of the 256 MB memory-side cache; the planes "
                                         "of a launch are 1.2 GB and are read by every tile row / column that needs them)",
                         "avg_launch_ms": avg_s * 1e3, "launches": gemm_cnt,
                         "note": "achieved = algorithmic int8 TOP/s: 36 digit pairs x 2 x rows x c x 2l per product over the "
                                 "summed launch time of the GEMM kernel; issued = the int8 operations the kernel executes "
                                 "(18 residue products instead of 36 digit-pair products when the residue formulation runs); "
                                 "gemm_phase = the same algorithmic work over operand split + GEMM + recombination; "
                                 "step_frac = the algorithmic int8 operations of a whole step (3 products) over ms_per_step"},
            "kernel_ms_per_step": breakdown,
        }
        if bitwise is not None:
            res["bitwise"] = merge_bitwise(bitwise)
        res.update(extras)
        if world == 1 and not args.no_cpu:
            res["cpu_baseline"] = cpu_baseline()
            if "bitwise" in res and "error" not in res["bitwise"]:
                res["bitwise"]["cpu_baseline"] = cpu_bitwise()
            if "config4" in res and "error" not in res["config4"]:
                res["config4"]["cpu_baseline"] = cpu_mlp()
                if "mlp" in res.get("host_engine", {}) and "bench_securenn_mlp" in res["host_engine"]["mlp"]:
                    res["config4"]["host_latency_ms"] = res["host_engine"]["mlp"]["bench_securenn_mlp"]["ms"]
            if "prg" in res and "error" not in res["prg"]:
                res["prg"]["cpu_baseline"] = cpu_prg()
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        tables.cleanup()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
