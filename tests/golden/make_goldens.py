"""Regenerates tests/golden/oracle_goldens.json from the CPU oracle.

These vectors are outputs of the oracle itself (oracle/mpc_oracle.cpp), not of the reference: the
reference engines cannot be built in the development image (Eigen 3.4.0 is absent).  They pin
regressions of the oracle and give the GPU tests fixed expected checksums for every protocol message;
what pins the oracle to the reference is listed in tests/test_oracle_kat.py.

Each entry is the FNV-1a-64 checksum (over little-endian 64-bit words; shorter payloads are
zero-padded to a multiple of 8 bytes) of every message a party sends and of every final share."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import my_protocols as mp  # noqa: E402
import protocols as pr  # noqa: E402
from my_backends import OracleMy  # noqa: E402
from oracle_lib import cks, prg_pops  # noqa: E402


def log_cks(w):
    return [[tag, cks(m)] for tag, m in w.log]


def pc_inputs(n, seed=22):
    w = prg_pops(seed, 0, 67 * n).reshape(n, 67)
    x, r = w[:, 0].copy(), w[:, 1].copy()
    r[2::5] = -1
    beta = (w[:, 2].view(np.uint64) % 2).astype(np.int8)
    s0 = ((w[:, 3:].view(np.uint64) & 0xFFFF) % 67).astype(np.uint16)
    bits = ((x.view(np.uint64)[:, None] >> np.arange(64, dtype=np.uint64)) & 1).astype(np.uint16)
    s1 = ((bits + 67 - s0) % 67).astype(np.uint16)
    return x, r, beta, np.ascontiguousarray(s0).ravel(), np.ascontiguousarray(s1).ravel()


def trunc_div2(a):
    return (a // 2 + ((a < 0) & (a % 2 != 0))).astype(np.int64)


def build(be):
    g = {}
    n = 64
    wd = prg_pops(20, 0, 2 * n * n)
    X, Y = wd[: n * n].copy(), wd[n * n:].copy()
    w = pr.World(be)
    Zs = pr.aby3_matmul(w, pr.aby3_share(w, X, n * n), pr.aby3_share(w, Y, n * n), n, n, n)
    g["aby3_matmul_64"] = dict(Z_share0=[cks(Zs[j][0]) for j in range(3)], Z_plain=cks(pr.aby3_reveal(w, Zs)[0]),
                               log=log_cks(w))
    w = pr.World(be)
    Zs = pr.snn_matmul(w, pr.snn_share(w, X, n * n), pr.snn_share(w, Y, n * n), n, n, n)
    g["snn_matmul_64"] = dict(Z_share=[cks(Zs[j]) for j in range(2)], Z_plain=cks(pr.snn_reveal(w, Zs)), log=log_cks(w))

    n = 1000
    a = trunc_div2(prg_pops(22, 67 * n, n))
    w = pr.World(be)
    Bs = pr.snn_msb(w, pr.snn_share(w, a, n)[:2])
    g["snn_msb_1000"] = dict(B_share=[cks(Bs[j]) for j in range(2)], log=log_cks(w))

    n = 500
    x, r, beta, s0, s1 = pc_inputs(n)
    w = pr.World(be)
    b2 = pr.snn_private_compare(w, [s0, s1], beta, r)
    g["snn_pc_500"] = dict(beta2=cks(b2), log=log_cks(w))

    n = 1000
    Xv = prg_pops(21, 0, n)
    Xv[:6] = [0, -1, 1, -2 ** 63, 2 ** 63 - 1, -2]
    w = pr.World(be)
    Xs = pr.aby3_share(w, Xv, n)
    Gs, Ps = pr.aby3_add_prepare(w, Xs)
    Zb = pr.aby3_msb(w, Gs, Ps)
    C = pr.aby3_bit_inject(w, Xs, Zb)
    g["aby3_relu_1000"] = dict(msb_share0=[cks(Zb[j][0]) for j in range(3)], C_share0=[cks(C[j][0]) for j in range(3)],
                               C_share1=[cks(C[j][1]) for j in range(3)], log=log_cks(w))

    # MyEngine: 64x64 product of the config-1 inputs, and msb + bit_inject at the reference benchmark's n = 2000
    n = 64
    w = pr.World(be)
    Zs = mp.my_matmul(w, mp.my_share(w, X, n * n), mp.my_share(w, Y, n * n), n, n, n)
    g["my_matmul_64"] = dict(Z_share=[cks(Zs[j]) for j in range(2)], Z_plain=cks(mp.my_reveal(w, Zs)), log=log_cks(w))
    n = 2000
    Xv = prg_pops(93, 0, n) // 2
    Xv[:5] = [0, -1, -2, -(1 << 62), (1 << 62) - 1]
    w = pr.World(be)
    Xs = mp.my_share(w, Xv, n)
    A = mp.my_msb(w, Xs)
    Zi = mp.my_bit_inject(w, Xs, A)
    g["my_relu_2000"] = dict(msb_share=[cks(A[j]) for j in range(2)], Z_share=[cks(Zi[j]) for j in range(2)], log=log_cks(w))
    return g


if __name__ == "__main__":
    out = os.path.join(HERE, "oracle_goldens.json")
    json.dump(build(OracleMy()), open(out, "w"), indent=1)
    print("wrote", out)
