"""Two interchangeable executors of the per-party local segments, used by the protocol drivers:

* OracleBackend -- the CPU restatement in oracle/ (the checker), defined here;
* GpuBackend / ResidentBackend -- the product (mpc_amd/backends.py): the C-ABI library running HIP kernels.

Both expose the same method names (the C-ABI entry points without their prefix) and take / return
numpy arrays, so a protocol written once runs on either and every message and share can be compared
byte for byte."""
import numpy as np

from mpc_amd.backends import GpuBackend, ResidentBackend, _c  # noqa: F401
from oracle_lib import i8, i64, oracle, u16


class OracleBackend:
    name = "oracle"

    def __init__(self):
        self.o = oracle()

    # ---- PRG
    def prg_fill_pop(self, seed, first, count, lcg=0):
        out = i64(count)
        self.o.orc_prg_fill_pop(seed, lcg, first, count, out)
        return out

    def prg_fill_buff(self, seed, first, count, lcg=0):
        a, b = i64(count), i8(count)
        self.o.orc_prg_fill_buff(seed, lcg, first, count, a, b)
        return a, b

    # ---- ring
    def ring_gemm(self, A, B, r, l, c, layA=0, layB=0, layC=0, C=None):
        acc = 0 if C is None else 1
        out = i64(r * c) if C is None else _c(C, np.int64).copy()
        self.o.orc_ring_gemm_u64(_c(A, np.int64), _c(B, np.int64), out, r, l, c, layA, layB, layC, acc)
        return out

    def ring_add(self, a, b):
        out = i64(a.size)
        self.o.orc_ring_add(_c(a, np.int64), _c(b, np.int64), out, a.size)
        return out

    def ring_sub(self, a, b):
        out = i64(a.size)
        self.o.orc_ring_sub(_c(a, np.int64), _c(b, np.int64), out, a.size)
        return out

    def ring_sar(self, a, k):
        out = i64(a.size)
        self.o.orc_ring_sar(_c(a, np.int64), out, a.size, k)
        return out

    def ring_shl1(self, a):
        out = i64(a.size)
        self.o.orc_ring_shl1(_c(a, np.int64), out, a.size)
        return out

    # ---- ABY3
    def aby3_share_local(self, X, n, seed_next, pop_next, seed_prev, pop_prev):
        out = i64(n)
        self.o.orc_aby3_share_local(_c(X, np.int64), n, seed_next, pop_next, seed_prev, pop_prev, out)
        return out

    def aby3_share_bools_local(self, X, n, seed_next, pop_next, seed_prev, pop_prev):
        out = i8(n)
        self.o.orc_aby3_share_bools_local(_c(X, np.int8), n, seed_next, pop_next, seed_prev, pop_prev, out)
        return out

    def aby3_matmul_local(self, X0, X1, Y0, Y1, r, l, c, layout, seed_next, pop_next, seed_prev, pop_prev):
        out = i64(r * c)
        self.o.orc_aby3_matmul_local(_c(X0, np.int64), _c(X1, np.int64), _c(Y0, np.int64), _c(Y1, np.int64), r, l, c,
                                     layout, seed_next, pop_next, seed_prev, pop_prev, out)
        return out

    def aby3_vecmul_local(self, X0, X1, Y0, Y1, n, seed_next, pop_next, seed_prev, pop_prev):
        out = i64(n)
        self.o.orc_aby3_vecmul_local(_c(X0, np.int64), _c(X1, np.int64), _c(Y0, np.int64), _c(Y1, np.int64), n,
                                     seed_next, pop_next, seed_prev, pop_prev, out)
        return out

    def aby3_reveal3(self, a, b, c):
        out = i64(a.size)
        self.o.orc_aby3_reveal3(_c(a, np.int64), _c(b, np.int64), _c(c, np.int64), out, a.size)
        return out

    def aby3_reveal3_bools(self, a, b, c):
        out = i8(a.size)
        self.o.orc_aby3_reveal3_bools(_c(a, np.int8), _c(b, np.int8), _c(c, np.int8), out, a.size)
        return out

    def aby3_reveal3_bits(self, a, b, c):
        out = i64(a.size)
        self.o.orc_aby3_reveal3_bits(_c(a, np.int64), _c(b, np.int64), _c(c, np.int64), out, a.size)
        return out

    def aby3_vecbool_mul_local(self, X0, X1, Y0, Y1):
        out = i8(X0.size)
        self.o.orc_aby3_vecbool_mul_local(_c(X0, np.int8), _c(X1, np.int8), _c(Y0, np.int8), _c(Y1, np.int8), X0.size, out)
        return out

    def aby3_add_prepare_step1(self, X0, X1):
        out = i64(X0.size)
        self.o.orc_aby3_add_prepare_step1(_c(X0, np.int64), _c(X1, np.int64), X0.size, out)
        return out

    def aby3_add_prepare_step2(self, X0, X1, C0, C1):
        n = X0.size
        P0, P1, G0 = i64(n), i64(n), i64(n)
        self.o.orc_aby3_add_prepare_step2(_c(X0, np.int64), _c(X1, np.int64), _c(C0, np.int64), _c(C1, np.int64), n,
                                          P0, P1, G0)
        return P0, P1, G0

    class _Msb:
        def __init__(self, o, G0, G1, P0, P1):
            import ctypes
            self.o, self.n = o, G0.size
            self.h = ctypes.c_void_p(o.raw("orc_aby3_msb_begin", ctypes.c_void_p)(
                _c(G0, np.int64), _c(G1, np.int64), _c(P0, np.int64), _c(P1, np.int64), G0.size))

        def round_pre(self, temp):
            msg = i8(self.n * 2 * temp)
            self.o.orc_aby3_msb_round_pre(self.h, temp, msg)
            return msg

        def round_post(self, temp, msg_in):
            self.o.orc_aby3_msb_round_post(self.h, temp, _c(msg_in, np.int8))

        def final(self):
            Z0, Z1 = i8(self.n), i8(self.n)
            self.o.orc_aby3_msb_final(self.h, Z0, Z1)
            return Z0, Z1

    def aby3_msb_begin(self, G0, G1, P0, P1):
        return OracleBackend._Msb(self.o, G0, G1, P0, P1)

    def aby3_trunc_local(self, party, X0, X1, number, seed12, pop12):
        out = i64(X0.size)
        self.o.orc_aby3_trunc_local(party, _c(X0, np.int64), _c(X1, np.int64), X0.size, number, seed12, pop12, out)
        return out

    def aby3_bit_inject_p0_pre(self, A0, B0, B1, seed01, pop01, seed20, pop20, rng=None):
        n = A0.size
        assert rng is None or tuple(rng) == (n, 0), "the CPU restatement handles whole vectors only"
        to1, to2 = i64(2 * n), i64(3 * n)
        self.o.orc_aby3_bit_inject_p0_pre(_c(A0, np.int64), _c(B0, np.int8), _c(B1, np.int8), n, seed01, pop01, seed20,
                                          pop20, to1, to2)
        return to1, to2

    def aby3_bit_inject_p0_post(self, B0, seed01, pop01, seed20, pop20, from1, from2, rng=None):
        n = B0.size
        assert rng is None or tuple(rng) == (n, 0), "the CPU restatement handles whole vectors only"
        C0, C1 = i64(n), i64(n)
        self.o.orc_aby3_bit_inject_p0_post(_c(B0, np.int8), n, seed01, pop01, seed20, pop20, _c(from1, np.int64),
                                           _c(from2, np.int64), C0, C1)
        return C0, C1

    def aby3_bit_inject_p1_pre(self, A0, A1, B0, B1, seed01, pop01, seed12, pop12, rng=None):
        n = A0.size
        assert rng is None or tuple(rng) == (n, 0), "the CPU restatement handles whole vectors only"
        to0, to2 = i64(2 * n), i64(3 * n)
        self.o.orc_aby3_bit_inject_p1_pre(_c(A0, np.int64), _c(A1, np.int64), _c(B0, np.int8), _c(B1, np.int8), n,
                                          seed01, pop01, seed12, pop12, to0, to2)
        return to0, to2

    def aby3_bit_inject_p1_post(self, B1, seed01, pop01, seed12, pop12, from0, from2, rng=None):
        n = B1.size
        assert rng is None or tuple(rng) == (n, 0), "the CPU restatement handles whole vectors only"
        C0, C1 = i64(n), i64(n)
        self.o.orc_aby3_bit_inject_p1_post(_c(B1, np.int8), n, seed01, pop01, seed12, pop12, _c(from0, np.int64),
                                           _c(from2, np.int64), C0, C1)
        return C0, C1

    def aby3_bit_inject_p2_pre(self, B0, B1, seed20, pop20, seed12, pop12, rng=None):
        n = B0.size
        assert rng is None or tuple(rng) == (n, 0), "the CPU restatement handles whole vectors only"
        to0, to1 = i64(n), i64(n)
        self.o.orc_aby3_bit_inject_p2_pre(_c(B0, np.int8), _c(B1, np.int8), n, seed20, pop20, seed12, pop12, to0, to1)
        return to0, to1

    def aby3_bit_inject_p2_post(self, B0, B1, seed20, pop20, seed12, pop12, from0, from1, rng=None):
        n = B0.size
        assert rng is None or tuple(rng) == (n, 0), "the CPU restatement handles whole vectors only"
        C0, C1 = i64(n), i64(n)
        self.o.orc_aby3_bit_inject_p2_post(_c(B0, np.int8), _c(B1, np.int8), n, seed20, pop20, seed12, pop12,
                                           _c(from0, np.int64), _c(from1, np.int64), C0, C1)
        return C0, C1

    # ---- SecureNN
    def snn_share_local(self, party, X, n, seed01, pop01):
        out = i64(n)
        self.o.orc_snn_share_local(party, _c(X, np.int64), n, seed01, pop01, out)
        return out

    def snn_add_L1(self, a, b):
        out = i64(a.size)
        self.o.orc_snn_add_L1(_c(a, np.int64), _c(b, np.int64), out, a.size)
        return out

    def snn_add_zp(self, a, b):
        out = u16(a.size)
        self.o.orc_snn_add_zp(_c(a, np.uint16), _c(b, np.uint16), out, a.size)
        return out

    def snn_beaver_dealer_matmul(self, seedS, r, l, c):
        sz = r * l + l * c + r * c
        m0, m1 = i64(sz), i64(sz)
        self.o.orc_snn_beaver_dealer_matmul(seedS, r, l, c, m0, m1)
        return m0, m1

    def snn_beaver_party_open_matmul(self, X, Y, layout, triple, r, l, c):
        ef = i64(r * l + l * c)
        self.o.orc_snn_beaver_party_open_matmul(_c(X, np.int64), _c(Y, np.int64), layout, _c(triple, np.int64), r, l,
                                                c, ef)
        return ef

    def snn_beaver_party_finish_matmul(self, party, triple, ef_own, ef_peer, r, l, c, layoutZ):
        Z = i64(r * c)
        self.o.orc_snn_beaver_party_finish_matmul(party, _c(triple, np.int64), _c(ef_own, np.int64),
                                                  _c(ef_peer, np.int64), r, l, c, layoutZ, Z)
        return Z

    def snn_beaver_dealer_vecmul(self, seedS, n_total, i0, n):
        o = [i64(n) for _ in range(6)]
        self.o.orc_snn_beaver_dealer_vecmul(seedS, n_total, i0, n, *o)
        return o  # A0,B0,C0,A1,B1,C1

    def snn_beaver_party_open_vecmul(self, X, Y, A, B):
        n = X.size
        E, F = i64(n), i64(n)
        self.o.orc_snn_beaver_party_open_vecmul(_c(X, np.int64), _c(Y, np.int64), _c(A, np.int64), _c(B, np.int64), n,
                                                E, F)
        return E, F

    def snn_beaver_party_finish_vecmul(self, party, A, B, C, E_own, F_own, E_peer, F_peer):
        n = A.size
        Z = i64(n)
        self.o.orc_snn_beaver_party_finish_vecmul(party, *[_c(x, np.int64) for x in
                                                           (A, B, C, E_own, F_own, E_peer, F_peer)], n, Z)
        return Z

    def snn_pc_party(self, party, xbits, beta, r, n_total, i0, seed01):
        n = beta.size
        d = u16(64 * n)
        self.o.orc_snn_pc_party(party, _c(xbits, np.uint16), _c(beta, np.int8), _c(r, np.int64), n_total, i0, n,
                                seed01, d)
        return d

    def snn_pc_helper(self, d0, d1):
        n = d0.size // 64
        b = i8(n)
        self.o.orc_snn_pc_helper(_c(d0, np.uint16), _c(d1, np.uint16), n, b)
        return b

    def snn_sc_party_open(self, party, A, n_total, i0, seed01):
        x = i64(A.size)
        self.o.orc_snn_sc_party_open(party, _c(A, np.int64), n_total, i0, A.size, seed01, x)
        return x

    def snn_sc_party_pc_inputs(self, n_total, i0, n, seed01):
        e, r = i8(n), i64(n)
        self.o.orc_snn_sc_party_pc_inputs(n_total, i0, n, seed01, e, r)
        return e, r

    def snn_sc_party_finish(self, party, A, delta_j, eta1_j, n_total, i0, seed01):
        y = i64(A.size)
        self.o.orc_snn_sc_party_finish(party, _c(A, np.int64), _c(delta_j, np.int64), _c(eta1_j, np.int64), n_total, i0,
                                       A.size, seed01, y)
        return y

    def snn_sc_helper_split(self, x0, x1, n_total, i0, seedS):
        n = x0.size
        xb0, d0, xb1, d1 = u16(64 * n), i64(n), u16(64 * n), i64(n)
        self.o.orc_snn_sc_helper_split(_c(x0, np.int64), _c(x1, np.int64), n_total, i0, n, seedS, xb0, d0, xb1, d1)
        return xb0, d0, xb1, d1

    def snn_sc_helper_eta(self, eta1_plain, n_total, i0, seedS):
        n = eta1_plain.size
        e0, e1 = i64(n), i64(n)
        self.o.orc_snn_sc_helper_eta(_c(eta1_plain, np.int8), n_total, i0, n, seedS, e0, e1)
        return e0, e1

    def snn_msbl1_helper_deal(self, n_total, i0, n, seedS):
        x0, l0, b0, x1, l1, b1 = i64(n), i64(n), u16(64 * n), i64(n), i64(n), u16(64 * n)
        self.o.orc_snn_msbl1_helper_deal(n_total, i0, n, seedS, x0, l0, b0, x1, l1, b1)
        return x0, l0, b0, x1, l1, b1

    def snn_msbl1_helper_beta(self, beta1_plain, n_total, i0, seedS):
        n = beta1_plain.size
        b0, b1 = i64(n), i64(n)
        self.o.orc_snn_msbl1_helper_beta(_c(beta1_plain, np.int8), n_total, i0, n, seedS, b0, b1)
        return b0, b1

    def snn_msbl1_party_open(self, A, xj):
        r = i64(A.size)
        self.o.orc_snn_msbl1_party_open(_c(A, np.int64), _c(xj, np.int64), A.size, r)
        return r

    def snn_msbl1_party_pc_inputs(self, r_own, r_peer, i0, seed01):
        n = r_own.size
        b, R = i8(n), i64(n)
        self.o.orc_snn_msbl1_party_pc_inputs(_c(r_own, np.int64), _c(r_peer, np.int64), i0, n, seed01, b, R)
        return b, R

    def snn_msbl1_party_mid(self, party, beta1_j, xlsb_j, beta0, R):
        n = R.size
        g, d = i64(n), i64(n)
        self.o.orc_snn_msbl1_party_mid(party, _c(beta1_j, np.int64), _c(xlsb_j, np.int64), _c(beta0, np.int8),
                                       _c(R, np.int64), n, g, d)
        return g, d

    def snn_msbl1_party_finish(self, gamma, delta, theta):
        n = gamma.size
        B = i64(n)
        self.o.orc_snn_msbl1_party_finish(_c(gamma, np.int64), _c(delta, np.int64), _c(theta, np.int64), n, B)
        return B
