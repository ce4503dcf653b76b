"""MyEngine segments (mpc_engine/MyEngine.cpp) on the two executors of tests/backends.py: the CPU oracle and
the HIP library.  Same method names on both, numpy in / numpy out."""
import numpy as np

from backends import OracleBackend, _c
from mpc_amd.my_backends import GpuMy  # noqa: F401
from oracle_lib import i8, i64, u16


class OracleMy(OracleBackend):
    def my_beaver_dealer_matmul(self, seed0, seed1, seedS, r, l, c):
        C0, C1 = i64(r * c), i64(r * c)
        self.o.orc_my_beaver_dealer_matmul(seed0, seed1, seedS, r, l, c, C0, C1)
        return C0, C1

    def my_beaver_party_open_matmul(self, X, Y, layout, seed2, r, l, c):
        ef = i64(r * l + l * c)
        self.o.orc_my_beaver_party_open_matmul(_c(X, np.int64), _c(Y, np.int64), layout, seed2, r, l, c, ef)
        return ef

    def my_beaver_party_finish_matmul(self, party, seed2, C, ef_own, ef_peer, r, l, c, layoutZ):
        Z = i64(r * c)
        self.o.orc_my_beaver_party_finish_matmul(party, seed2, _c(C, np.int64), _c(ef_own, np.int64),
                                                 _c(ef_peer, np.int64), r, l, c, layoutZ, Z)
        return Z

    def my_beaver_dealer_vecmul(self, seed0, seed1, seedS, n_total, i0, n):
        C0, C1 = i64(n), i64(n)
        self.o.orc_my_beaver_dealer_vecmul(seed0, seed1, seedS, n_total, i0, n, C0, C1)
        return C0, C1

    def my_beaver_party_open_vecmul(self, X, Y, seed2, n_total, i0):
        n = X.size
        ef = i64(2 * n)
        self.o.orc_my_beaver_party_open_vecmul(_c(X, np.int64), _c(Y, np.int64), seed2, n_total, i0, n, ef)
        return ef

    def my_beaver_party_finish_vecmul(self, party, seed2, C, ef_own, ef_peer, n_total, i0):
        n = C.size
        Z = i64(n)
        self.o.orc_my_beaver_party_finish_vecmul(party, seed2, _c(C, np.int64), _c(ef_own, np.int64),
                                                 _c(ef_peer, np.int64), n_total, i0, n, Z)
        return Z

    def my_pc_party(self, party, xbits, beta, r, i0, seed01):
        n = beta.size
        d = u16(64 * n)
        self.o.orc_my_pc_party(party, _c(xbits, np.uint16), _c(beta, np.int8), _c(r, np.int64), i0, n, seed01, d)
        return d

    def my_msb_helper_deal(self, seed0, seed1, seedS, i0, n):
        l0, b0, l1, b1 = i8(n), u16(64 * n), i8(n), u16(64 * n)
        self.o.orc_my_msb_helper_deal(seed0, seed1, seedS, i0, n, l0, b0, l1, b1)
        return l0, b0, l1, b1

    def my_msb_helper_beta(self, beta1_plain, n_total, i0, seedS):
        n = beta1_plain.size
        b0, b1 = i8(n), i8(n)
        self.o.orc_my_msb_helper_beta(_c(beta1_plain, np.int8), n_total, i0, n, seedS, b0, b1)
        return b0, b1

    def my_msb_party_open(self, A, seed2, i0):
        r = i64(A.size)
        self.o.orc_my_msb_party_open(_c(A, np.int64), seed2, i0, A.size, r)
        return r

    def my_msb_party_pc_inputs(self, r_own, r_peer, i0, seed01):
        n = r_own.size
        b, R = i8(n), i64(n)
        self.o.orc_my_msb_party_pc_inputs(_c(r_own, np.int64), _c(r_peer, np.int64), i0, n, seed01, b, R)
        return b, R

    def my_msb_party_finish(self, party, beta1_j, beta0, xlsb_j, R):
        n = R.size
        B = i8(n)
        self.o.orc_my_msb_party_finish(party, _c(beta1_j, np.int8), _c(beta0, np.int8), _c(xlsb_j, np.int8),
                                       _c(R, np.int64), n, B)
        return B

    def my_bit_inject_dealer(self, seed0, seed1, seedS, n_total, i0, n):
        o = [i64(n) for _ in range(8)]
        self.o.orc_my_bit_inject_dealer(seed0, seed1, seedS, n_total, i0, n, *o)
        return o  # V0,W0,S0,T0,V1,W1,S1,T1

    def my_bit_inject_party_pre(self, X, A, seed2, i0):
        n = X.size
        y, c = i64(n), i8(n)
        self.o.orc_my_bit_inject_party_pre(_c(X, np.int64), _c(A, np.int8), seed2, i0, n, y, c)
        return y, c

    def my_bit_inject_party_post(self, party, seed2, i0, y_own, c_own, y_peer, c_peer, V, W, S, T):
        n = y_own.size
        Z = i64(n)
        self.o.orc_my_bit_inject_party_post(party, seed2, i0, n, _c(y_own, np.int64), _c(c_own, np.int8),
                                            _c(y_peer, np.int64), _c(c_peer, np.int8), _c(V, np.int64), _c(W, np.int64),
                                            _c(S, np.int64), _c(T, np.int64), Z)
        return Z

    def xor_bools(self, a, b):
        out = i8(a.size)
        self.o.orc_xor_bools(_c(a, np.int8), _c(b, np.int8), out, a.size)
        return out
