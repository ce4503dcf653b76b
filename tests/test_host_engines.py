"""The C++ host engines (mpc_amd/host): drop-in ABY3Engine / SecurennEngine classes over the C ABI,
the in-process messenger and the wire-compatible SocketMessenger."""
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "mpc_amd", "host")
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_goldens.json")))


@pytest.mark.parametrize("mib", [32, 256])
def test_socket_messenger_large_rounds_do_not_deadlock(mib):
    """32 and 256 MiB per peer per round over loopback TCP (the reference's blocking messenger hangs above ~1 MiB;
    a 4096 x 4096 share is 128 MiB).  The same program passes with 2 GiB payloads, the 16384 x 16384 message size
    (20 s and 25 GB of memory: run by hand as `mpc_amd/host/socket_selftest 2147483648`)."""
    exe = os.path.join(HOST, "socket_selftest")
    if not os.path.exists(exe):
        pytest.skip("host tools not built")
    r = subprocess.run([exe, str(mib << 20)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok" in r.stdout


def test_wire_codecs_round_trip_and_byte_positions():
    """test/test1.cpp:97-155, :230-262: every codec gives its input back; plus the byte positions of the wire layouts."""
    exe = os.path.join(HOST, "codec_selftest")
    if not os.path.exists(exe):
        pytest.skip("host tools not built")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr


def test_in_process_messenger_and_host_copy():
    """Three party threads, sixty rounds, payloads from empty to 3 MiB through InProcMessenger and the library's
    multi-threaded host copy (no GPU involved): every received message is what its sender posted."""
    exe = os.path.join(HOST, "inproc_selftest")
    if not os.path.exists(exe):
        pytest.skip("host tools not built")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("rows,cols", [(1, 1), (3, 5), (31, 33), (257, 256), (1000, 777), (2048, 1031)])
def test_host_transpose_matches_numpy(rows, cols):
    """mpcx_host_transpose64 (the codecs' row-major <-> column-major conversion; needs no GPU): single-threaded below
    2^16 words, in bands over the library's host threads above; ragged shapes leave partial tiles on both sides."""
    import ctypes
    import numpy as np
    import mpc_amd
    lib = mpc_amd.load_library()
    src = np.random.default_rng(rows * cols).integers(-2 ** 63, 2 ** 63, (rows, cols), dtype=np.int64)
    dst = np.full((cols, rows), -1, dtype=np.int64)
    U64 = ctypes.c_uint64
    assert lib.mpcx_host_transpose64(ctypes.c_void_p(dst.ctypes.data), ctypes.c_void_p(src.ctypes.data), U64(rows), U64(cols)) == 0
    assert np.array_equal(dst, src.T)
    assert lib.mpcx_host_transpose64(ctypes.c_void_p(src.ctypes.data), ctypes.c_void_p(src.ctypes.data), U64(rows), U64(cols)) != 0


def test_host_headers_keep_the_reference_surface():
    """Same class / method / member names as mpc_engine/ABY3Engine.hpp:24-54 and SecurennEngine.hpp:20-50."""
    a = open(os.path.join(HOST, "mpc_engine", "ABY3Engine.hpp")).read()
    for name in ("share", "share_matrix", "share_bools", "reveal", "reveal_matrix", "reveal_bools", "reveal_bits",
                 "matrix_mul", "vector_mul", "vecbool_mul", "add_prepare", "msb", "trunc", "bit_inject"):
        assert re.search(r"void %s\(" % name, a), name
    s = open(os.path.join(HOST, "mpc_engine", "SecurennEngine.hpp")).read()
    for name in ("matrix_mul", "vector_mul", "reveal", "reveal_matrix", "reveal_vec_bools", "reveal_L_1", "reveal_zp",
                 "share", "privateCompare", "shareConvert", "msb_L_1", "msb"):
        assert re.search(r"void %s\(" % name, s), name
    for hdr in (a, s):
        for member in ("VirtuleMessenger *messenger", "size_t party_id", "vector<PRgenerator> prgs",
                       "vector<string> message_send", "vector<string> message_recv"):
            assert member in hdr


def test_host_surface_compiles_to_the_reference_signatures():
    """mpc_amd/host/test/surface_check.cpp states, as static_asserts on pointer-to-member types, every public method,
    data member, alias, constant and codec of the reference's headers with its exact signature, and the idioms the
    reference's drivers use on an engine (prgs[k].pop_int64(), int_buff, party_id).  Syntax-only compile, nothing is
    linked or run; the negative control (one deliberately wrong signature) must fail."""
    import shutil
    if not shutil.which("g++"):
        pytest.skip("no host compiler")
    src = os.path.join(HOST, "test", "surface_check.cpp")
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wno-sign-compare", src]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run(cmd + ["-DSURFACE_NEGATIVE_CONTROL"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "static assertion failed" in r.stderr


@pytest.mark.gpu
def test_host_engines_selftest_and_known_answers(orc):
    """Three party threads, in-process messenger, every engine method; share checksums equal the oracle's.  The
    engines' relu (one protocol, device-resident between its rounds) is compared inside the program with the calls it
    is made of, word for word and generator position for generator position, and here with the oracle's chain."""
    exe = os.path.join(HOST, "host_selftest")
    assert os.path.exists(exe), "mpc_amd/host/host_selftest missing: run __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    kat = {}
    for line in r.stdout.splitlines():
        m = re.match(r"KAT (\S+) (\d) ([0-9A-F]{16})", line)
        if m:
            kat[(m.group(1), int(m.group(2)))] = m.group(3)
    g = GOLD["aby3_matmul_64"]
    assert [kat[("aby3_matmul_64_Z0", j)] for j in range(3)] == g["Z_share0"]
    assert {kat[("aby3_matmul_64_Zplain", j)] for j in range(3)} == {g["Z_plain"]}
    assert [kat[("snn_msb_1000_B", j)] for j in range(2)] == GOLD["snn_msb_1000"]["B_share"]
    # ABY3Engine::relu on 3001 values of PRgenerator(21,0) (the first five replaced by corner values) against the
    # oracle's share -> add_prepare -> msb -> bit_inject -> subtract
    import protocols as pr
    from oracle_lib import fnv1a64, prg_pops
    X = prg_pops(21, 0, 3001)
    X[:5] = [0, -1, 1, -2 ** 63, 2 ** 63 - 1]
    w = pr.World(orc)
    Xs = pr.aby3_share(w, X, X.size)
    Gs, Ps = pr.aby3_add_prepare(w, Xs)
    C = pr.aby3_bit_inject(w, Xs, pr.aby3_msb(w, Gs, Ps))
    for j in range(3):
        assert kat[("aby3_relu_fused_3001_R0", j)] == "%016X" % fnv1a64(orc.ring_sub(Xs[j][0], C[j][0]))
    # SecurennEngine::relu on 2001 values pop / 2 of PRgenerator(23,0) (the first three replaced)
    A = prg_pops(23, 0, 2001)
    A = (A + (A < 0)) >> 1  # pop / 2 in C++ truncates toward zero
    A[:3] = [0, -1, 1]
    w = pr.World(orc)
    As = pr.snn_share(w, A, A.size)[:2]
    M = pr.snn_vecmul(w, As, pr.snn_msb(w, As))
    for j in range(2):
        assert kat[("snn_relu_fused_2001_R", j)] == "%016X" % fnv1a64(orc.ring_sub(As[j], M[j]))


@pytest.mark.gpu
def test_host_engines_with_protocol_self_checks():
    """The same program over engines compiled with -DMYDEBUG (the reference's switch, mpc_engine/ABY3Engine.cpp:3):
    every product, add_prepare, msb, bit_inject, shareConvert and msb_L_1 opens its operands and its result and asserts
    the relation between them; the share checksums are the same as without the checks (openings draw no randomness)."""
    exe = os.path.join(HOST, "host_selftest_checked")
    assert os.path.exists(exe), "mpc_amd/host/host_selftest_checked missing: run __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    plain = subprocess.run([os.path.join(HOST, "host_selftest")], capture_output=True, text=True, timeout=300)
    kat = lambda out: sorted(l for l in out.splitlines() if l.startswith("KAT "))
    assert kat(r.stdout) == kat(plain.stdout) and kat(r.stdout)


@pytest.mark.gpu
def test_three_processes_over_socket_messenger():
    """The reference's own process model (benchmark/makefile:13-17: `./benchmark1 0 &`, `1 &`, `2`): three single-party
    processes over the wire-compatible SocketMessenger (127.0.0.1, ports 8000-8002), every engine on the one GPU.  Each
    process checks every result against the plain computation."""
    exe = os.path.join(HOST, "benchmark1")
    assert os.path.exists(exe), "mpc_amd/host/benchmark1 missing: run __graft_entry__.build()"
    args = ["--width", "96", "--vec", "3000", "--repeat", "1"]
    procs = [subprocess.Popen([exe, str(p)] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for p in range(3)]
    outs = []
    try:
        for pr_ in procs:
            outs.append(pr_.communicate(timeout=240)[0])
    finally:
        for pr_ in procs:
            if pr_.poll() is None:
                pr_.kill()
    for p, (pr_, out) in enumerate(zip(procs, outs)):
        assert pr_.returncode == 0 and "all results verified" in out, f"party {p}: {out[-2000:]}"
