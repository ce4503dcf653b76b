"""Several GPUs behind one party (mpcx_group_*): two contexts on the one GPU of the test box, driven through the
group entry points, reproduce the single-context results bit for bit."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from oracle_lib import KEY01, KEY20

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "mpc_amd", "host")
U64, I32 = ctypes.c_uint64, ctypes.c_int


@pytest.fixture(scope="module")
def lib():
    import mpc_amd
    L = mpc_amd.load_library()
    L.mpcx_group_last_error.restype = ctypes.c_char_p
    L.mpcx_group_ctx.restype = ctypes.c_void_p
    return L


def ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


@pytest.mark.parametrize("members", [1, 2, 3])
@pytest.mark.parametrize("r,l,c,layout", [(300, 70, 130, 0), (300, 70, 130, 1), (5, 4, 3, 1), (1000, 260, 520, 0),
                                          (2048, 1024, 512, 1)])  # the last: operands beyond the staged-copy threshold
def test_group_matmul_equals_the_oracle(lib, orc, members, r, l, c, layout):
    """mpcx_group_aby3_matmul_local on host operands: the row blocks of the members, each written into its place
    of Z0, are the whole product + mask of the CPU oracle (both operand layouts)."""
    g = ctypes.c_void_p()
    devs = (I32 * members)(*([0] * members))
    assert lib.mpcx_group_init(devs, I32(members), ctypes.byref(g)) == 0, lib.mpcx_last_error(None)
    try:
        assert lib.mpcx_group_size(g) == members
        rng = np.random.default_rng(r * 31 + members)
        X0, X1, Y0, Y1 = (rng.integers(-2 ** 63, 2 ** 63, k, dtype=np.int64) for k in (r * l, r * l, l * c, l * c))
        Z = np.zeros(r * c, dtype=np.int64)
        rc = lib.mpcx_group_aby3_matmul_local(g, ptr(X0), ptr(X1), ptr(Y0), ptr(Y1), U64(r), U64(l), U64(c), I32(layout),
                                              U64(KEY01), U64(11), U64(KEY20), U64(7), ptr(Z))
        assert rc == 0, lib.mpcx_group_last_error(g)
        assert np.array_equal(Z, orc.aby3_matmul_local(X0, X1, Y0, Y1, r, l, c, layout, KEY01, 11, KEY20, 7))
    finally:
        lib.mpcx_group_shutdown(g)


@pytest.mark.parametrize("members", [1, 3])
@pytest.mark.parametrize("r,l,c,layout", [(300, 70, 130, 0), (300, 70, 130, 1), (1000, 260, 520, 1)])
def test_group_matmul_with_a_resident_operand(lib, orc, members, r, l, c, layout):
    """mpcx_group_operand_upload keeps both shares of Y on every member; two products with different X and different
    stream offsets through mpcx_group_aby3_matmul_local_resident are the oracle's (and so the per-call form's); an
    operand of another shape is refused."""
    g = ctypes.c_void_p()
    assert lib.mpcx_group_init((I32 * members)(*([0] * members)), I32(members), ctypes.byref(g)) == 0
    try:
        rng = np.random.default_rng(r * 17 + members + layout)
        Y0, Y1 = (rng.integers(-2 ** 63, 2 ** 63, l * c, dtype=np.int64) for _ in range(2))
        op = ctypes.c_void_p()
        assert lib.mpcx_group_operand_upload(g, ptr(Y0), ptr(Y1), U64(l), U64(c), I32(layout), ctypes.byref(op)) == 0, \
            lib.mpcx_group_last_error(g)
        for pop in (11, 12345):
            X0, X1 = (rng.integers(-2 ** 63, 2 ** 63, r * l, dtype=np.int64) for _ in range(2))
            Z = np.zeros(r * c, dtype=np.int64)
            rc = lib.mpcx_group_aby3_matmul_local_resident(g, ptr(X0), ptr(X1), op, U64(r), U64(l), U64(c), I32(layout),
                                                           U64(KEY01), U64(pop), U64(KEY20), U64(pop + 3), ptr(Z))
            assert rc == 0, lib.mpcx_group_last_error(g)
            assert np.array_equal(Z, orc.aby3_matmul_local(X0, X1, Y0, Y1, r, l, c, layout, KEY01, pop, KEY20, pop + 3))
        Z = np.zeros(r * (c + 1), dtype=np.int64)
        assert lib.mpcx_group_aby3_matmul_local_resident(g, ptr(X0), ptr(X1), op, U64(r), U64(l), U64(c + 1), I32(layout),
                                                         U64(KEY01), U64(0), U64(KEY20), U64(0), ptr(Z)) != 0
        assert b"operand is" in lib.mpcx_group_last_error(g)
        assert lib.mpcx_group_operand_free(g, op) == 0
    finally:
        lib.mpcx_group_shutdown(g)


@pytest.mark.parametrize("members", [1, 3])
@pytest.mark.parametrize("party", [0, 1])
@pytest.mark.parametrize("r,l,c", [(300, 70, 130), (5, 4, 3), (700, 260, 128)])
def test_group_beaver_finish_equals_the_oracle(lib, orc, members, party, r, l, c):
    """mpcx_group_snn_beaver_party_finish_matmul on the host messages of SecurennEngine::matrix_mul: the members' row
    blocks, each written into its place of Z, are the oracle's Z = C + E B + A F + party E F."""
    g = ctypes.c_void_p()
    assert lib.mpcx_group_init((I32 * members)(*([0] * members)), I32(members), ctypes.byref(g)) == 0
    try:
        rng = np.random.default_rng(r + 7 * party + members)
        triple = rng.integers(-2 ** 63, 2 ** 63, r * l + l * c + r * c, dtype=np.int64)
        ef_own, ef_peer = (rng.integers(-2 ** 63, 2 ** 63, r * l + l * c, dtype=np.int64) for _ in range(2))
        Z = np.zeros(r * c, dtype=np.int64)
        rc = lib.mpcx_group_snn_beaver_party_finish_matmul(g, I32(party), ptr(triple), ptr(ef_own), ptr(ef_peer), U64(r),
                                                           U64(l), U64(c), ptr(Z))
        assert rc == 0, lib.mpcx_group_last_error(g)
        assert np.array_equal(Z, orc.snn_beaver_party_finish_matmul(party, triple, ef_own, ef_peer, r, l, c, 0))
    finally:
        lib.mpcx_group_shutdown(g)


@pytest.mark.parametrize("members", [1, 2, 3])
@pytest.mark.parametrize("r,l,c", [(300, 70, 130), (5, 4, 3), (700, 260, 128)])
def test_group_beaver_dealer_equals_the_oracle(lib, orc, members, r, l, c):
    """mpcx_group_snn_beaver_dealer_matmul: the members deal the rows of A_i and C_i of their row blocks (the first one
    B_i as well) straight into the two host messages, which are the oracle's A_i | B_i | C_i."""
    from oracle_lib import KEYS
    g = ctypes.c_void_p()
    assert lib.mpcx_group_init((I32 * members)(*([0] * members)), I32(members), ctypes.byref(g)) == 0
    try:
        size = r * l + l * c + r * c
        m0, m1 = np.full(size, -1, dtype=np.int64), np.full(size, -1, dtype=np.int64)
        rc = lib.mpcx_group_snn_beaver_dealer_matmul(g, U64(KEYS), U64(r), U64(l), U64(c), ptr(m0), ptr(m1))
        assert rc == 0, lib.mpcx_group_last_error(g)
        w0, w1 = orc.snn_beaver_dealer_matmul(KEYS, r, l, c)
        assert np.array_equal(m0, w0) and np.array_equal(m1, w1)
    finally:
        lib.mpcx_group_shutdown(g)


@pytest.mark.parametrize("members", [1, 3])
@pytest.mark.parametrize("r,l,c", [(300, 70, 130), (5, 4, 3), (700, 260, 128)])
def test_group_my_engine_matmul_equals_the_oracle(lib, members, r, l, c):
    """MyEngine's matrix_mul over a group: the dealer's C_0 / C_1 and, for both parties, Z = C + E B + A F + party E F in
    row blocks written into place equal the oracle's whole-matrix segments."""
    from my_backends import OracleMy
    from oracle_lib import KEY12, KEYS
    omy = OracleMy()
    g = ctypes.c_void_p()
    assert lib.mpcx_group_init((I32 * members)(*([0] * members)), I32(members), ctypes.byref(g)) == 0
    try:
        C0, C1 = np.full(r * c, -1, dtype=np.int64), np.full(r * c, -1, dtype=np.int64)
        rc = lib.mpcx_group_my_beaver_dealer_matmul(g, U64(KEY20), U64(KEY12), U64(KEYS), U64(r), U64(l), U64(c), ptr(C0), ptr(C1))
        assert rc == 0, lib.mpcx_group_last_error(g)
        w0, w1 = omy.my_beaver_dealer_matmul(KEY20, KEY12, KEYS, r, l, c)
        assert np.array_equal(C0, w0) and np.array_equal(C1, w1)
        rng = np.random.default_rng(r + members)
        for party, seed2, C in ((0, KEY20, C0), (1, KEY12, C1)):
            ef_own, ef_peer = (rng.integers(-2 ** 63, 2 ** 63, r * l + l * c, dtype=np.int64) for _ in range(2))
            Z = np.zeros(r * c, dtype=np.int64)
            rc = lib.mpcx_group_my_beaver_party_finish_matmul(g, I32(party), U64(seed2), ptr(C), ptr(ef_own), ptr(ef_peer),
                                                              U64(r), U64(l), U64(c), ptr(Z))
            assert rc == 0, lib.mpcx_group_last_error(g)
            assert np.array_equal(Z, omy.my_beaver_party_finish_matmul(party, seed2, C, ef_own, ef_peer, r, l, c, 0))
    finally:
        lib.mpcx_group_shutdown(g)


def test_group_for_each_ranges(lib):
    """The planner cuts [0, n) into contiguous multiples of the granule, and every member's callback sees its range."""
    g = ctypes.c_void_p()
    assert lib.mpcx_group_init((I32 * 3)(0, 0, 0), I32(3), ctypes.byref(g)) == 0
    seen = []
    FN = ctypes.CFUNCTYPE(I32, ctypes.c_void_p, I32, U64, U64, ctypes.c_void_p)

    def cb(ctx, member, first, count, user):
        seen.append((member, first, count, ctx == lib.mpcx_group_ctx(g, member)))
        return 0

    try:
        for n, gran in ((1000, 64), (64, 64), (1, 64), (130, 1), (0, 64)):
            seen.clear()
            assert lib.mpcx_group_for_each(g, U64(n), U64(gran), FN(cb), None) == 0
            got = sorted(seen)
            assert all(ok for *_, ok in got)
            assert got[0][1] == 0 and sum(c for _, _, c, _ in got) == n
            for (_, f0, c0, _), (_, f1, _, _) in zip(got, got[1:]):
                assert f0 + c0 == f1 and f1 % gran == 0
        failing = FN(lambda ctx, member, first, count, user: 1 if member == 1 else 0)
        assert lib.mpcx_group_for_each(g, U64(1000), U64(64), failing, None) != 0
    finally:
        lib.mpcx_group_shutdown(g)


def kat_lines(env):
    exe = os.path.join(HOST, "host_selftest")
    assert os.path.exists(exe), "mpc_amd/host/host_selftest missing: run __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return sorted(l for l in r.stdout.splitlines() if re.match(r"KAT ", l))


def test_host_engines_over_a_group_give_the_same_shares():
    """The C++ engines' own test program with MPCX_DEVICES=0,0 (ABY3 matrix_mul in row blocks, add_prepare / msb /
    bit_inject in element ranges; SecurennEngine vector_mul / privateCompare / shareConvert / msb_L_1 / msb and MyEngine
    vector_mul / privateCompare / msb / bit_inject in element ranges over two and three contexts): every assertion holds and every share checksum is the one of the
    single-context run."""
    one = kat_lines({"MPCX_DEVICES": ""})
    two = kat_lines({"MPCX_DEVICES": "0,0"})
    three = kat_lines({"MPCX_DEVICES": "0,0,0"})
    assert one and one == two == three


def test_host_benchmark_over_a_group_verifies():
    """benchmark1's ABY3 lines (512 x 512 matrix_mul + trunc in row blocks, a 100 000-element ReLU chain in ranges)
    over two contexts: the program checks every result against the plain computation."""
    exe = os.path.join(HOST, "benchmark1")
    assert os.path.exists(exe), "mpc_amd/host/benchmark1 missing: run __graft_entry__.build()"
    r = subprocess.run([exe, "--engine", "aby3", "--width", "512", "--vec", "100000", "--repeat", "1", "--no-mlp"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, MPCX_DEVICES="0,0"))
    assert r.returncode == 0 and "all results verified" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("pin", [False, True])
def test_host_benchmark_over_a_group_with_staged_operands(pin):
    """The same at 2048 x 2048 (operands of 32 MiB: every upload goes through the pinned staging buffers, a member's row
    block of the column-major X is gathered by the host threads), three party threads with a group of two each, with
    the second operand uploaded per call and resident on the members (ABY3Engine::pin)."""
    exe = os.path.join(HOST, "benchmark1")
    assert os.path.exists(exe), "mpc_amd/host/benchmark1 missing: run __graft_entry__.build()"
    args = [exe, "--engine", "aby3", "--width", "2048", "--vec", "4096", "--repeat", "1", "--no-mlp"] + (["--pin"] if pin else [])
    r = subprocess.run(args, capture_output=True, text=True, timeout=180, env=dict(os.environ, MPCX_DEVICES="0,0"))
    assert r.returncode == 0 and "all results verified" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("engine", ["securenn", "my"])
def test_host_additive_engines_benchmark_over_a_group_verifies(engine):
    """benchmark1's SecureNN / MyEngine lines (msb + vector_mul / bit_inject on 100 000 elements in ranges over three
    contexts, matrix_mul in row blocks over them): the program checks every result against the plain computation."""
    exe = os.path.join(HOST, "benchmark1")
    assert os.path.exists(exe), "mpc_amd/host/benchmark1 missing: run __graft_entry__.build()"
    r = subprocess.run([exe, "--engine", engine, "--width", "256", "--vec", "100000", "--repeat", "1", "--no-mlp"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, MPCX_DEVICES="0,0,0"))
    assert r.returncode == 0 and "all results verified" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_checkpoint_tables_are_shared_through_the_disk_cache(tmp_path):
    """MPCX_CKPT_CACHE: a process leaves the checkpoint table of a stream it walked in the directory; the next process
    loads it (and produces the same words); a truncated or foreign file is ignored."""
    import sys
    prog = ("import sys, numpy as np; sys.path.insert(0, %r); from mpc_amd import Context; c = Context(0); "
            "d = c.empty(np.int64, 1000); c.call('prg_fill_pop', 0x77, 0, 3_000_000, 1000, d); c.sync(); "
            "print('WORDS', ' '.join('%%x' %% (int(v) & (2**64 - 1)) for v in d.to_host()[:8]))" % ROOT)
    env = dict(os.environ, MPCX_CKPT_CACHE=str(tmp_path), MPCX_PRG_WALK_AHEAD="0")
    run = lambda e: subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300, env=e)
    first = run(env)
    assert first.returncode == 0, first.stderr[-2000:]
    files = list(tmp_path.glob("prg_*.ckpt"))
    assert len(files) == 1 and files[0].stat().st_size > 24 * (3_000_000 // 64)
    second = run(env)
    assert second.returncode == 0 and second.stdout == first.stdout and "WORDS" in first.stdout
    no_cache = run(dict(os.environ, MPCX_CKPT_CACHE=""))
    assert no_cache.stdout == first.stdout
    data = files[0].read_bytes()
    files[0].write_bytes(data[: len(data) // 2])  # truncated: ignored, walked again and replaced
    third = run(env)
    assert third.returncode == 0 and third.stdout == first.stdout
    assert files[0].stat().st_size == len(data)
    files[0].write_bytes(data[:40] + bytes(len(data) - 40))  # right header, wrong states: rejected
    fourth = run(env)
    assert fourth.returncode == 0 and fourth.stdout == first.stdout
    assert files[0].read_bytes() == data  # ... and replaced by the table walked again
    # one state damaged far behind the first entry, exactly where the words asked for are replayed from: the checksum
    # of the body rejects the file (accepted, it would give other words)
    at = 40 + 24 * ((320005 + 3_000_000) // 64) + 3
    files[0].write_bytes(data[:at] + bytes([data[at] ^ 0x10]) + data[at + 1:])
    fifth = run(env)
    assert fifth.returncode == 0 and fifth.stdout == first.stdout
    # a header that promises 2^35 entries: the length of the file does not match, nothing is allocated for it
    files[0].write_bytes(data[:24] + (1 << 35).to_bytes(8, "little") + data[32:])
    sixth = run(env)
    assert sixth.returncode == 0 and sixth.stdout == first.stdout


def test_checkpoint_cache_default_directory(tmp_path):
    """Without MPCX_CKPT_CACHE the tables go to $XDG_CACHE_HOME/mpcx (the directory is created); an empty
    MPCX_CKPT_CACHE switches the cache off."""
    import sys
    prog = ("import sys, numpy as np; sys.path.insert(0, %r); from mpc_amd import Context; c = Context(0); "
            "c.call('prg_ckpt_build', 0x78, 0, 2_000_000, 0); c.sync()" % ROOT)
    env = {k: v for k, v in os.environ.items() if k != "MPCX_CKPT_CACHE"}
    env["XDG_CACHE_HOME"] = str(tmp_path / "xdg")
    os.makedirs(env["XDG_CACHE_HOME"])
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(list((tmp_path / "xdg" / "mpcx").glob("prg_*.ckpt"))) == 1
    env["XDG_CACHE_HOME"] = str(tmp_path / "xdg2")
    os.makedirs(env["XDG_CACHE_HOME"])
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300,
                       env=dict(env, MPCX_CKPT_CACHE=""))
    assert r.returncode == 0 and not (tmp_path / "xdg2" / "mpcx").exists()


def test_sparse_marks_parallel_walk_is_exact_and_not_trusted(tmp_path):
    """The library walks a long stream on many host threads between the sparse marks shipped beside it
    (mpc_amd/prg_marks.bin).  The words far into the KEY01 stream (buffer index 2^27, stream word 2^28) are the oracle's
    with the marks, without them (MPCX_PRG_MARKS empty: the serial walk), and with a marks file in which one state was
    damaged (a segment then does not end where the next mark says, and the serial walk takes over)."""
    import sys
    from oracle_lib import prg_buff
    first, count = (1 << 27) - 500, 1000
    prog = ("import sys, time, numpy as np; sys.path.insert(0, %r); from mpc_amd import Context; c = Context(0); "
            "t0 = time.perf_counter(); c.call('prg_ckpt_build', 0xAB01, 0, 0, %d); dt = time.perf_counter() - t0; "
            "d = c.empty(np.int64, %d); b = c.empty(np.int8, %d); c.call('prg_fill_buff', 0xAB01, 0, %d, %d, d, b); c.sync(); "
            "print('WALK_S', dt); print('WORDS', ' '.join('%%x' %% (int(v) & (2**64 - 1)) for v in d.to_host()), "
            "''.join(str(int(v)) for v in b.to_host()))" % (ROOT, first + count, count, count, first, count))
    base = dict(os.environ, MPCX_CKPT_CACHE="", MPCX_PRG_WALK_AHEAD="0")
    run = lambda e: subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300, env=e)
    words = lambda r: [l for l in r.stdout.splitlines() if l.startswith("WORDS")][0]
    secs = lambda r: float([l for l in r.stdout.splitlines() if l.startswith("WALK_S")][0].split()[1])
    with_marks = run({k: v for k, v in base.items() if k != "MPCX_PRG_MARKS"})
    assert with_marks.returncode == 0, with_marks.stderr[-2000:]
    ints, bools = prg_buff(0xAB01, first, count)
    want = "WORDS " + " ".join("%x" % (int(v) & (2 ** 64 - 1)) for v in ints) + " " + "".join(str(int(v)) for v in bools)
    assert words(with_marks) == want
    serial = run(dict(base, MPCX_PRG_MARKS=""))
    assert serial.returncode == 0 and words(serial) == want
    data = bytearray(open(os.path.join(ROOT, "mpc_amd", "prg_marks.bin"), "rb").read())
    data[24 + 24 + 24 * 10 + 5] ^= 0x40  # the first table is KEY01's: its mark 10
    bad = tmp_path / "marks.bin"
    bad.write_bytes(bytes(data))
    damaged = run(dict(base, MPCX_PRG_MARKS=str(bad)))
    assert damaged.returncode == 0 and words(damaged) == want
    assert secs(with_marks) < secs(serial), (secs(with_marks), secs(serial))


def test_stream_consumed_while_the_background_thread_extends_it(tmp_path):
    """A pop stream read further and further (what an engine does): from the second request on the library's background
    thread walks ahead of the reader, appends to the table, mirrors it on the device (in place, and into larger copies
    when it outgrows one) and writes it to the cache directory, holding the table's lock for a slice at a time.  Every
    request returns the oracle's words whatever the thread is doing at that moment, and the file it leaves behind is one
    the next process accepts."""
    import sys
    from oracle_lib import prg_pops
    step, rounds, count = 5_000_000, 14, 512
    prog = ("import sys, time, glob, numpy as np; sys.path.insert(0, %r); from mpc_amd import Context; c = Context(0); "
            "d = c.empty(np.int64, %d)\n"
            "for i in range(%d):\n"
            "    c.call('prg_fill_pop', 0x79, 0, i * %d, %d, d); c.sync()\n"
            "    print('WORDS', i, ' '.join('%%x' %% (int(v) & (2**64 - 1)) for v in d.to_host()))\n"
            "    time.sleep(0.02 * (i %% 3))\n"
            "t0 = time.time()\n"
            "while time.time() - t0 < 20 and not glob.glob(%r + '/prg_*.ckpt'): time.sleep(0.05)\n"
            "c.close()\n" % (ROOT, count, rounds, step, count, str(tmp_path)))
    env = dict(os.environ, MPCX_CKPT_CACHE=str(tmp_path))
    env.pop("MPCX_PRG_WALK_AHEAD", None)
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l.split() for l in r.stdout.splitlines() if l.startswith("WORDS")]
    assert len(lines) == rounds
    for l in lines:
        i = int(l[1])
        want = ["%x" % (int(v) & (2 ** 64 - 1)) for v in prg_pops(0x79, i * step, count)]
        assert l[2:] == want, f"request {i} differs from the oracle"
    files = list(tmp_path.glob("prg_*.ckpt"))
    assert len(files) == 1 and not list(tmp_path.glob("*.tmp.*"))
    size = files[0].stat().st_size
    assert size >= 40 + 24 * (2 * step // 64)  # written by the background thread, after it walked ahead at least once
    # the next process (background thread off) accepts the file: it reads words near its end without rewriting it
    last = (size - 40) // 24 * 64 - 320000 - 8192  # pops start at stream word 320000 (utils/PRgenerator.hpp:6)
    prog2 = ("import sys, numpy as np; sys.path.insert(0, %r); from mpc_amd import Context; c = Context(0); "
             "d = c.empty(np.int64, %d); c.call('prg_fill_pop', 0x79, 0, %d, %d, d); c.sync(); "
             "print('WORDS', ' '.join('%%x' %% (int(v) & (2**64 - 1)) for v in d.to_host()))" % (ROOT, count, last - count, count))
    mtime = files[0].stat().st_mtime_ns
    r2 = subprocess.run([sys.executable, "-c", prog2], capture_output=True, text=True, timeout=300,
                        env=dict(env, MPCX_PRG_WALK_AHEAD="0"))
    assert r2.returncode == 0, r2.stderr[-2000:]
    want = "WORDS " + " ".join("%x" % (int(v) & (2 ** 64 - 1)) for v in prg_pops(0x79, last - count, count))
    assert r2.stdout.strip() == want
    assert files[0].stat().st_mtime_ns == mtime and files[0].stat().st_size == size


def test_three_contexts_read_one_growing_stream_concurrently():
    """Three host threads, each with a context of its own on the same GPU (the three parties of a single-process run),
    read one generator stream at interleaved, advancing positions while the background thread extends its table: the
    table's host copy moves when it grows, its device copy is replaced while kernels of the other contexts may still
    read the old one, and every thread still gets the oracle's words."""
    import threading
    from mpc_amd import Context
    from oracle_lib import prg_pops
    seed, step, rounds, count = 0x7A, 1_700_000, 18, 384
    results, errors = {}, []

    def party(k):
        try:
            ctx = Context(0)
            d = ctx.empty(np.int64, count)
            for i in range(rounds):
                first = (3 * i + k) * step
                ctx.call("prg_fill_pop", seed, 0, first, count, d)
                ctx.sync()
                results[(k, i)] = d.to_host().copy()
            ctx.close()
        except Exception as ex:  # surfaced by the main thread
            errors.append(f"thread {k}: {ex}")

    th = [threading.Thread(target=party, args=(k,)) for k in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    for (k, i), got in sorted(results.items()):
        assert np.array_equal(got, prg_pops(seed, (3 * i + k) * step, count)), f"thread {k}, request {i}"


@pytest.mark.parametrize("nbytes", [1, 4095, (4 << 20) - 1, 4 << 20, (4 << 20) + 1, (12 << 20) - 3, (37 << 20) + 5])
def test_host_device_copies_round_trip(nbytes):
    """mpcx_h2d / mpcx_d2h (plain below 4 MiB, pipelined through the pinned staging buffers above, ragged last chunks)
    and mpcx_host_copy (the multi-threaded memcpy) return exactly the bytes they were given."""
    from mpc_amd import Context, load_library
    ctx = Context(0)
    rng = np.random.default_rng(nbytes)
    src = rng.integers(0, 256, nbytes + 9, dtype=np.uint8)[9:]  # an odd host alignment as well
    d = ctx.to_device(src)
    assert np.array_equal(d.to_host(), src)
    # device to device through a second buffer, then back: the staged download of a buffer the staged upload filled
    e = ctx.to_device(np.zeros(nbytes, np.uint8))
    ctx.call("d2d", e, d, nbytes)
    assert np.array_equal(e.to_host(), src)
    dst = np.zeros(nbytes, np.uint8)
    lib = load_library()
    assert lib.mpcx_host_copy(ctypes.c_void_p(dst.ctypes.data), ctypes.c_void_p(src.ctypes.data), U64(nbytes)) == 0
    assert np.array_equal(dst, src)
    ctx.close()
