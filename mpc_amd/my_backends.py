"""MyEngine segments (mpc_engine/MyEngine.cpp) on the GPU backend (the C-ABI entry points mpcx_my_*)."""
import numpy as np

from .backends import GpuBackend, _c


class GpuMy(GpuBackend):
    def my_beaver_dealer_matmul(self, seed0, seed1, seedS, r, l, c):
        return self._go("my_beaver_dealer_matmul", seed0, seed1, seedS, r, l, c, ("out", np.int64, r * c),
                        ("out", np.int64, r * c))

    def my_beaver_party_open_matmul(self, X, Y, layout, seed2, r, l, c):
        return self._go("my_beaver_party_open_matmul", _c(X, np.int64), _c(Y, np.int64), layout, seed2, r, l, c,
                        ("out", np.int64, r * l + l * c))

    def my_beaver_party_finish_matmul(self, party, seed2, C, ef_own, ef_peer, r, l, c, layoutZ):
        return self._go("my_beaver_party_finish_matmul", party, seed2, _c(C, np.int64), _c(ef_own, np.int64),
                        _c(ef_peer, np.int64), r, l, c, layoutZ, ("out", np.int64, r * c))

    def my_beaver_dealer_vecmul(self, seed0, seed1, seedS, n_total, i0, n):
        return self._go("my_beaver_dealer_vecmul", seed0, seed1, seedS, n_total, i0, n, ("out", np.int64, n),
                        ("out", np.int64, n))

    def my_beaver_party_open_vecmul(self, X, Y, seed2, n_total, i0):
        n = X.size
        return self._go("my_beaver_party_open_vecmul", _c(X, np.int64), _c(Y, np.int64), seed2, n_total, i0, n,
                        ("out", np.int64, 2 * n))

    def my_beaver_party_finish_vecmul(self, party, seed2, C, ef_own, ef_peer, n_total, i0):
        n = C.size
        return self._go("my_beaver_party_finish_vecmul", party, seed2, _c(C, np.int64), _c(ef_own, np.int64),
                        _c(ef_peer, np.int64), n_total, i0, n, ("out", np.int64, n))

    def my_pc_party(self, party, xbits, beta, r, i0, seed01):
        n = beta.size
        return self._go("my_pc_party", party, _c(xbits, np.uint16), _c(beta, np.int8), _c(r, np.int64), i0, n, seed01,
                        ("out", np.uint16, 64 * n))

    def my_msb_helper_deal(self, seed0, seed1, seedS, i0, n):
        return self._go("my_msb_helper_deal", seed0, seed1, seedS, i0, n, ("out", np.int8, n), ("out", np.uint16, 64 * n),
                        ("out", np.int8, n), ("out", np.uint16, 64 * n))

    def my_msb_helper_beta(self, beta1_plain, n_total, i0, seedS):
        n = beta1_plain.size
        return self._go("my_msb_helper_beta", _c(beta1_plain, np.int8), n_total, i0, n, seedS, ("out", np.int8, n),
                        ("out", np.int8, n))

    def my_msb_party_open(self, A, seed2, i0):
        return self._go("my_msb_party_open", _c(A, np.int64), seed2, i0, A.size, ("out", np.int64, A.size))

    def my_msb_party_pc_inputs(self, r_own, r_peer, i0, seed01):
        n = r_own.size
        return self._go("my_msb_party_pc_inputs", _c(r_own, np.int64), _c(r_peer, np.int64), i0, n, seed01,
                        ("out", np.int8, n), ("out", np.int64, n))

    def my_msb_party_finish(self, party, beta1_j, beta0, xlsb_j, R):
        n = R.size
        return self._go("my_msb_party_finish", party, _c(beta1_j, np.int8), _c(beta0, np.int8), _c(xlsb_j, np.int8),
                        _c(R, np.int64), n, ("out", np.int8, n))

    def my_bit_inject_dealer(self, seed0, seed1, seedS, n_total, i0, n):
        return self._go("my_bit_inject_dealer", seed0, seed1, seedS, n_total, i0, n, *[("out", np.int64, n)] * 8)

    def my_bit_inject_party_pre(self, X, A, seed2, i0):
        n = X.size
        return self._go("my_bit_inject_party_pre", _c(X, np.int64), _c(A, np.int8), seed2, i0, n, ("out", np.int64, n),
                        ("out", np.int8, n))

    def my_bit_inject_party_post(self, party, seed2, i0, y_own, c_own, y_peer, c_peer, V, W, S, T):
        n = y_own.size
        return self._go("my_bit_inject_party_post", party, seed2, i0, n, _c(y_own, np.int64), _c(c_own, np.int8),
                        _c(y_peer, np.int64), _c(c_peer, np.int8), _c(V, np.int64), _c(W, np.int64), _c(S, np.int64),
                        _c(T, np.int64), ("out", np.int64, n))

    def xor_bools(self, a, b):
        return self._go("xor_bools", _c(a, np.int8), _c(b, np.int8), ("out", np.int8, a.size), a.size)
