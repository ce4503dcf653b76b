"""Executors of the per-party local segments on the product: libmpcx.so running HIP kernels on one GPU.

* GpuBackend      -- host arrays in and out (uploaded / downloaded around each call; test-sized inputs);
* ResidentBackend -- arrays stay on the device (borrowed from torch's allocator), nothing synchronises: whole
  protocols with every message handed over as a device pointer (bench.py).

The method names are the C-ABI entry points without their prefix; mpc_amd/protocols.py composes them into the
three-party protocols.  There is no fallback of any kind: without the library or a GPU these classes fail."""
import numpy as np


def _c(a, dt):
    if a is None or hasattr(a, "ptr"):  # None, or a buffer already on the device
        return a
    return np.ascontiguousarray(a, dtype=dt)


class GpuBackend:
    """Same surface as OracleBackend, executed by libmpcx.so on cuda:0 (no fallback of any kind)."""
    name = "gpu"
    bit_inject_post_takes_minuend = True  # aby3_bit_inject_p*_post(..., sub=(S0, S1)) -> mpcx_aby3_bit_inject_p*_post_sub

    def __init__(self, ctx=None):
        from mpc_amd import Context
        self.ctx = ctx or Context(0)

    def _up(self, a, dt):
        return None if a is None else self.ctx.to_device(a, dt)

    def _go(self, name, *args):
        """args: numpy arrays are inputs (uploaded), ('out', dtype, n) tuples are outputs, the rest scalars."""
        ctx = self.ctx
        dev, outs = [], []
        for a in args:
            if isinstance(a, tuple) and len(a) == 3 and a[0] == "out":
                d = ctx.empty(a[1], a[2])
                outs.append(d)
                dev.append(d)
            elif isinstance(a, np.ndarray):
                dev.append(ctx.to_device(a))
            else:
                dev.append(a)
        ctx.call(name, *dev)
        ctx.sync()
        res = [o.to_host() for o in outs]
        return res[0] if len(res) == 1 else res

    # ---- PRG
    def prg_fill_pop(self, seed, first, count, lcg=0):
        return self._go("prg_fill_pop", seed, lcg, first, count, ("out", np.int64, count))

    def prg_fill_buff(self, seed, first, count, lcg=0):
        return self._go("prg_fill_buff", seed, lcg, first, count, ("out", np.int64, count), ("out", np.int8, count))

    # ---- ring
    def ring_gemm(self, A, B, r, l, c, layA=0, layB=0, layC=0, C=None):
        ctx = self.ctx
        dA, dB = ctx.to_device(A, np.int64), ctx.to_device(B, np.int64)
        dC = ctx.empty(np.int64, r * c) if C is None else ctx.to_device(C, np.int64)
        ctx.call("ring_gemm_u64", dA, dB, dC, r, l, c, layA, layB, layC, 0 if C is None else 1)
        ctx.sync()
        return dC.to_host()

    def ring_add(self, a, b):
        return self._go("ring_add", _c(a, np.int64), _c(b, np.int64), ("out", np.int64, a.size), a.size)

    def ring_sub(self, a, b):
        return self._go("ring_sub", _c(a, np.int64), _c(b, np.int64), ("out", np.int64, a.size), a.size)

    def ring_sar(self, a, k):
        return self._go("ring_sar", _c(a, np.int64), ("out", np.int64, a.size), a.size, k)

    def ring_shl1(self, a):
        return self._go("ring_shl1", _c(a, np.int64), ("out", np.int64, a.size), a.size)

    def relayout(self, a, rows, cols, ls, ld):
        return self._go("relayout", _c(a, np.int64), ("out", np.int64, a.size), rows, cols, ls, ld)

    # ---- ABY3
    def aby3_share_local(self, X, n, seed_next, pop_next, seed_prev, pop_prev):
        return self._go("aby3_share_local", _c(X, np.int64), n, seed_next, pop_next, seed_prev, pop_prev,
                        ("out", np.int64, n))

    def aby3_share_bools_local(self, X, n, seed_next, pop_next, seed_prev, pop_prev):
        return self._go("aby3_share_bools_local", _c(X, np.int8), n, seed_next, pop_next, seed_prev, pop_prev,
                        ("out", np.int8, n))

    def aby3_matmul_local(self, X0, X1, Y0, Y1, r, l, c, layout, seed_next, pop_next, seed_prev, pop_prev):
        return self._go("aby3_matmul_local", _c(X0, np.int64), _c(X1, np.int64), _c(Y0, np.int64), _c(Y1, np.int64), r,
                        l, c, layout, seed_next, pop_next, seed_prev, pop_prev, ("out", np.int64, r * c))

    def aby3_matmul_local_rows(self, X0, X1, Y0, Y1, rows, l, c, row0, seed_next, pop_next, seed_prev, pop_prev):
        return self._go("aby3_matmul_local_rows", _c(X0, np.int64), _c(X1, np.int64), _c(Y0, np.int64),
                        _c(Y1, np.int64), rows, l, c, row0, seed_next, pop_next, seed_prev, pop_prev,
                        ("out", np.int64, rows * c))

    def aby3_vecmul_local(self, X0, X1, Y0, Y1, n, seed_next, pop_next, seed_prev, pop_prev):
        return self._go("aby3_vecmul_local", _c(X0, np.int64), _c(X1, np.int64), _c(Y0, np.int64), _c(Y1, np.int64), n,
                        seed_next, pop_next, seed_prev, pop_prev, ("out", np.int64, n))

    def aby3_reveal3(self, a, b, c):
        return self._go("aby3_reveal3", _c(a, np.int64), _c(b, np.int64), _c(c, np.int64), ("out", np.int64, a.size),
                        a.size)

    def aby3_reveal3_bools(self, a, b, c):
        return self._go("aby3_reveal3_bools", _c(a, np.int8), _c(b, np.int8), _c(c, np.int8), ("out", np.int8, a.size),
                        a.size)

    def aby3_reveal3_bits(self, a, b, c):
        return self._go("aby3_reveal3_bits", _c(a, np.int64), _c(b, np.int64), _c(c, np.int64), ("out", np.int64, a.size),
                        a.size)

    def aby3_vecbool_mul_local(self, X0, X1, Y0, Y1):
        return self._go("aby3_vecbool_mul_local", _c(X0, np.int8), _c(X1, np.int8), _c(Y0, np.int8), _c(Y1, np.int8),
                        X0.size, ("out", np.int8, X0.size))

    def aby3_add_prepare_step1(self, X0, X1):
        return self._go("aby3_add_prepare_step1", _c(X0, np.int64), _c(X1, np.int64), X0.size,
                        ("out", np.int64, X0.size))

    def aby3_add_prepare_step2(self, X0, X1, C0, C1):
        n = X0.size
        return self._go("aby3_add_prepare_step2", _c(X0, np.int64), _c(X1, np.int64), _c(C0, np.int64),
                        _c(C1, np.int64), n, ("out", np.int64, n), ("out", np.int64, n), ("out", np.int64, n))

    msb_msg_skew = 0

    class _Msb:
        def __init__(self, ctx, G0, G1, P0, P1):
            self.ctx, self.n = ctx, G0.size
            self.P0, self.P1 = ctx.to_device(P0, np.int64), ctx.to_device(P1, np.int64)
            self.G0, self.G1 = ctx.to_device(G0, np.int64), ctx.to_device(G1, np.int64)  # read in place by round 32
            self.state = ctx.empty(np.uint64, 6 * self.n)
            ctx.call("aby3_msb_begin", self.G0, self.G1, self.P0, self.P1, self.n, self.state)
            ctx.sync()

        def round_pre(self, temp):
            k = GpuBackend.msb_msg_skew  # messages at this byte offset from an aligned allocation
            buf = self.ctx.empty(np.int8, self.n * 2 * temp + k)
            msg = buf.view(k, self.n * 2 * temp)
            self.ctx.call("aby3_msb_round_pre", self.state, self.n, temp, msg)
            self.ctx.sync()
            return msg.to_host()

        def round_post(self, temp, msg_in):
            k = GpuBackend.msb_msg_skew
            buf = self.ctx.to_device(np.concatenate([np.zeros(k, np.int8), np.asarray(msg_in, np.int8)]), np.int8)
            self.ctx.call("aby3_msb_round_post", self.state, self.n, temp, buf.view(k, msg_in.size))
            self.ctx.sync()

        def final(self):
            Z0, Z1 = self.ctx.empty(np.int8, self.n), self.ctx.empty(np.int8, self.n)
            self.ctx.call("aby3_msb_final", self.state, self.P0, self.P1, self.n, Z0, Z1)
            self.ctx.sync()
            return Z0.to_host(), Z1.to_host()

        def _skewed(self, host):
            k = GpuBackend.msb_msg_skew
            buf = self.ctx.to_device(np.concatenate([np.zeros(k, np.int8), np.asarray(host, np.int8)]), np.int8)
            return buf.view(k, host.size)

        def round_mid(self, temp, msg_in):
            k = GpuBackend.msb_msg_skew
            out = self.ctx.empty(np.int8, self.n * temp + k).view(k, self.n * temp)
            self.ctx.call("aby3_msb_round_mid", self.state, self.n, temp, self._skewed(msg_in), out)
            self.ctx.sync()
            return out.to_host()

        def last(self, msg_in):
            Z0, Z1 = self.ctx.empty(np.int8, self.n), self.ctx.empty(np.int8, self.n)
            self.ctx.call("aby3_msb_last", self.state, self.n, self._skewed(msg_in), Z0, Z1)
            self.ctx.sync()
            return Z0.to_host(), Z1.to_host()

    def aby3_msb_begin(self, G0, G1, P0, P1):
        return GpuBackend._Msb(self.ctx, G0, G1, P0, P1)

    def aby3_trunc_local(self, party, X0, X1, number, seed12, pop12):
        return self._go("aby3_trunc_local", party, _c(X0, np.int64), _c(X1, np.int64), X0.size, number, seed12, pop12,
                        ("out", np.int64, X0.size))

    # bit_inject: rng = (n_total, i0) when the arrays are elements [i0, i0 + n) of a longer vector
    def aby3_bit_inject_p0_pre(self, A0, B0, B1, seed01, pop01, seed20, pop20, rng=None):
        n = A0.size
        nt, i0 = rng or (n, 0)
        return self._go("aby3_bit_inject_p0_pre", _c(A0, np.int64), _c(B0, np.int8), _c(B1, np.int8), nt, i0, n, seed01,
                        pop01, seed20, pop20, ("out", np.int64, 2 * n), ("out", np.int64, 3 * n))

    # sub = (S0, S1): the post segment fused with the ReLU subtraction, returns S_k - C_k instead of C_k
    @staticmethod
    def _minuend(sub):
        return ("_post_sub", [_c(sub[0], np.int64), _c(sub[1], np.int64)]) if sub is not None else ("_post", [])

    def aby3_bit_inject_p0_post(self, B0, seed01, pop01, seed20, pop20, from1, from2, rng=None, sub=None):
        n = B0.size
        nt, i0 = rng or (n, 0)
        tail, S = self._minuend(sub)
        return self._go("aby3_bit_inject_p0" + tail, _c(B0, np.int8), nt, i0, n, seed01, pop01, seed20, pop20,
                        _c(from1, np.int64), _c(from2, np.int64), *S, ("out", np.int64, n), ("out", np.int64, n))

    def aby3_bit_inject_p1_pre(self, A0, A1, B0, B1, seed01, pop01, seed12, pop12, rng=None):
        n = A0.size
        nt, i0 = rng or (n, 0)
        return self._go("aby3_bit_inject_p1_pre", _c(A0, np.int64), _c(A1, np.int64), _c(B0, np.int8), _c(B1, np.int8),
                        nt, i0, n, seed01, pop01, seed12, pop12, ("out", np.int64, 2 * n), ("out", np.int64, 3 * n))

    def aby3_bit_inject_p1_post(self, B1, seed01, pop01, seed12, pop12, from0, from2, rng=None, sub=None):
        n = B1.size
        nt, i0 = rng or (n, 0)
        tail, S = self._minuend(sub)
        return self._go("aby3_bit_inject_p1" + tail, _c(B1, np.int8), nt, i0, n, seed01, pop01, seed12, pop12,
                        _c(from0, np.int64), _c(from2, np.int64), *S, ("out", np.int64, n), ("out", np.int64, n))

    def aby3_bit_inject_p2_pre(self, B0, B1, seed20, pop20, seed12, pop12, rng=None):
        n = B0.size
        nt, i0 = rng or (n, 0)
        return self._go("aby3_bit_inject_p2_pre", _c(B0, np.int8), _c(B1, np.int8), nt, i0, n, seed20, pop20, seed12, pop12,
                        ("out", np.int64, n), ("out", np.int64, n))

    def aby3_bit_inject_p2_post(self, B0, B1, seed20, pop20, seed12, pop12, from0, from1, rng=None, sub=None):
        n = B0.size
        nt, i0 = rng or (n, 0)
        tail, S = self._minuend(sub)
        return self._go("aby3_bit_inject_p2" + tail, _c(B0, np.int8), _c(B1, np.int8), nt, i0, n, seed20, pop20, seed12,
                        pop12, _c(from0, np.int64), _c(from1, np.int64), *S, ("out", np.int64, n), ("out", np.int64, n))

    # ---- SecureNN
    def snn_share_local(self, party, X, n, seed01, pop01):
        return self._go("snn_share_local", party, _c(X, np.int64), n, seed01, pop01, ("out", np.int64, n))

    def snn_add_L1(self, a, b):
        return self._go("snn_add_L1", _c(a, np.int64), _c(b, np.int64), ("out", np.int64, a.size), a.size)

    def snn_add_zp(self, a, b):
        return self._go("snn_add_zp", _c(a, np.uint16), _c(b, np.uint16), ("out", np.uint16, a.size), a.size)

    def snn_beaver_dealer_matmul(self, seedS, r, l, c):
        sz = r * l + l * c + r * c
        return self._go("snn_beaver_dealer_matmul", seedS, r, l, c, ("out", np.int64, sz), ("out", np.int64, sz))

    def snn_beaver_party_open_matmul(self, X, Y, layout, triple, r, l, c):
        return self._go("snn_beaver_party_open_matmul", _c(X, np.int64), _c(Y, np.int64), layout, _c(triple, np.int64),
                        r, l, c, ("out", np.int64, r * l + l * c))

    def snn_beaver_party_finish_matmul(self, party, triple, ef_own, ef_peer, r, l, c, layoutZ):
        return self._go("snn_beaver_party_finish_matmul", party, _c(triple, np.int64), _c(ef_own, np.int64),
                        _c(ef_peer, np.int64), r, l, c, layoutZ, ("out", np.int64, r * c))

    def snn_beaver_dealer_vecmul(self, seedS, n_total, i0, n):
        return self._go("snn_beaver_dealer_vecmul", seedS, n_total, i0, n, *[("out", np.int64, n)] * 6)

    def snn_beaver_party_open_vecmul(self, X, Y, A, B):
        n = X.size
        return self._go("snn_beaver_party_open_vecmul", _c(X, np.int64), _c(Y, np.int64), _c(A, np.int64),
                        _c(B, np.int64), n, ("out", np.int64, n), ("out", np.int64, n))

    def snn_beaver_party_finish_vecmul(self, party, A, B, C, E_own, F_own, E_peer, F_peer):
        n = A.size
        return self._go("snn_beaver_party_finish_vecmul", party,
                        *[_c(x, np.int64) for x in (A, B, C, E_own, F_own, E_peer, F_peer)], n, ("out", np.int64, n))

    def snn_pc_party(self, party, xbits, beta, r, n_total, i0, seed01):
        n = beta.size
        return self._go("snn_pc_party", party, _c(xbits, np.uint16), _c(beta, np.int8), _c(r, np.int64), n_total, i0, n,
                        seed01, ("out", np.uint16, 64 * n))

    def snn_pc_helper(self, d0, d1):
        n = d0.size // 64
        return self._go("snn_pc_helper", _c(d0, np.uint16), _c(d1, np.uint16), n, ("out", np.int8, n))

    def snn_sc_party_open(self, party, A, n_total, i0, seed01):
        return self._go("snn_sc_party_open", party, _c(A, np.int64), n_total, i0, A.size, seed01,
                        ("out", np.int64, A.size))

    def snn_sc_party_pc_inputs(self, n_total, i0, n, seed01):
        return self._go("snn_sc_party_pc_inputs", n_total, i0, n, seed01, ("out", np.int8, n), ("out", np.int64, n))

    def snn_sc_party_finish(self, party, A, delta_j, eta1_j, n_total, i0, seed01):
        return self._go("snn_sc_party_finish", party, _c(A, np.int64), _c(delta_j, np.int64), _c(eta1_j, np.int64),
                        n_total, i0, A.size, seed01, ("out", np.int64, A.size))

    def snn_sc_helper_split(self, x0, x1, n_total, i0, seedS):
        n = x0.size
        return self._go("snn_sc_helper_split", _c(x0, np.int64), _c(x1, np.int64), n_total, i0, n, seedS,
                        ("out", np.uint16, 64 * n), ("out", np.int64, n), ("out", np.uint16, 64 * n),
                        ("out", np.int64, n))

    def snn_sc_helper_eta(self, eta1_plain, n_total, i0, seedS):
        n = eta1_plain.size
        return self._go("snn_sc_helper_eta", _c(eta1_plain, np.int8), n_total, i0, n, seedS, ("out", np.int64, n),
                        ("out", np.int64, n))

    def snn_msbl1_helper_deal(self, n_total, i0, n, seedS):
        return self._go("snn_msbl1_helper_deal", n_total, i0, n, seedS, ("out", np.int64, n), ("out", np.int64, n),
                        ("out", np.uint16, 64 * n), ("out", np.int64, n), ("out", np.int64, n),
                        ("out", np.uint16, 64 * n))

    def snn_msbl1_helper_beta(self, beta1_plain, n_total, i0, seedS):
        n = beta1_plain.size
        return self._go("snn_msbl1_helper_beta", _c(beta1_plain, np.int8), n_total, i0, n, seedS, ("out", np.int64, n),
                        ("out", np.int64, n))

    def snn_msbl1_party_open(self, A, xj):
        return self._go("snn_msbl1_party_open", _c(A, np.int64), _c(xj, np.int64), A.size, ("out", np.int64, A.size))

    def snn_msbl1_party_pc_inputs(self, r_own, r_peer, i0, seed01):
        n = r_own.size
        return self._go("snn_msbl1_party_pc_inputs", _c(r_own, np.int64), _c(r_peer, np.int64), i0, n, seed01,
                        ("out", np.int8, n), ("out", np.int64, n))

    def snn_msbl1_party_mid(self, party, beta1_j, xlsb_j, beta0, R):
        n = R.size
        return self._go("snn_msbl1_party_mid", party, _c(beta1_j, np.int64), _c(xlsb_j, np.int64), _c(beta0, np.int8),
                        _c(R, np.int64), n, ("out", np.int64, n), ("out", np.int64, n))

    def snn_msbl1_party_finish(self, gamma, delta, theta):
        n = gamma.size
        return self._go("snn_msbl1_party_finish", _c(gamma, np.int64), _c(delta, np.int64), _c(theta, np.int64), n,
                        ("out", np.int64, n))


    # ---- SecureNN msb, one call per messenger gap
    def snn_msb_party_open(self, party, A, n_total, i0, seed01):
        return self._go("snn_msb_party_open", party, _c(A, np.int64), n_total, i0, A.size, seed01, ("out", np.int64, A.size))

    def snn_msb_party_convert_open(self, party, A, delta_j, eta1_j, xj, n_total, i0, seed01):
        return self._go("snn_msb_party_convert_open", party, _c(A, np.int64), _c(delta_j, np.int64), _c(eta1_j, np.int64),
                        _c(xj, np.int64), n_total, i0, A.size, seed01, ("out", np.int64, A.size))

    def snn_msb_party_pc_inputs(self, r_own, r_peer, i0, seed01):
        n = r_own.size
        return self._go("snn_msb_party_pc_inputs", _c(r_own, np.int64), _c(r_peer, np.int64), i0, n, seed01,
                        ("out", np.int8, n), ("out", np.int64, n))

    def snn_msb_party_mid_open(self, party, beta1_j, xlsb_j, beta0, R, A, B):
        n = R.size
        if not self.snn_msb_keeps_gamma_delta:  # finish forms gamma + delta again from E, F and the triple
            E, F = self._go("snn_msb_party_mid_open", party, _c(beta1_j, np.int64), _c(xlsb_j, np.int64),
                            _c(beta0, np.int8), _c(R, np.int64), _c(A, np.int64), _c(B, np.int64), n, None, None,
                            *[("out", np.int64, n)] * 2)
            return None, None, E, F
        return self._go("snn_msb_party_mid_open", party, _c(beta1_j, np.int64), _c(xlsb_j, np.int64), _c(beta0, np.int8),
                        _c(R, np.int64), _c(A, np.int64), _c(B, np.int64), n, *[("out", np.int64, n)] * 4)

    snn_msb_keeps_gamma_delta = False

    def snn_msb_party_finish(self, party, A, B, C, E_own, F_own, E_peer, F_peer, gamma, delta):
        n = A.size
        return self._go("snn_msb_party_finish", party,
                        *[_c(x, np.int64) for x in (A, B, C, E_own, F_own, E_peer, F_peer)],
                        None if gamma is None else _c(gamma, np.int64), None if delta is None else _c(delta, np.int64), n,
                        ("out", np.int64, n))


class ResidentBackend(GpuBackend):
    """GpuBackend whose arrays stay on the device: inputs and outputs are DeviceArrays borrowed from torch's
    caching allocator, nothing is copied to the host and nothing synchronises.  Used to time whole protocols
    (bench.py); `up` / `down` move test data in and out."""
    name = "gpu-resident"

    def __init__(self, ctx, torch, dev):
        super().__init__(ctx)
        self.torch, self.dev = torch, dev
        self._tdt = {np.dtype(np.int64): torch.int64, np.dtype(np.uint64): torch.int64, np.dtype(np.int8): torch.int8,
                     np.dtype(np.uint16): torch.int16, np.dtype(np.uint32): torch.int32}

    def empty(self, dtype, n):
        t = self.torch.empty(max(int(n), 1), dtype=self._tdt[np.dtype(dtype)], device=self.dev)
        return self.ctx.wrap(t, dtype).view(0, int(n))

    def up(self, a):
        a = np.ascontiguousarray(a)
        t = self.torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else a).to(self.dev)
        return self.ctx.wrap(t, a.dtype)

    def down(self, d):
        self.ctx.sync()
        return d.to_host()

    def tensor(self, d):
        """The device array as a torch tensor on the same memory (no copy, no synchronisation)."""
        t = d
        while not hasattr(t, "data_ptr"):
            t = t._owner
        off = d.ptr - t.data_ptr()
        flat = t.view(-1).view(self.torch.uint8)[off:off + d.nbytes]
        return flat.view(self._tdt[d.dtype])

    def _go(self, name, *args):
        dev, outs = [], []
        for a in args:
            if isinstance(a, tuple) and len(a) == 3 and a[0] == "out":
                d = self.empty(a[1], a[2])
                outs.append(d)
                dev.append(d)
            else:
                dev.append(a)
        self._run(name, *dev)
        return outs[0] if len(outs) == 1 else outs

    def _run(self, name, *dev):
        if self.events is None:
            self.ctx.call(name, *dev)
            return
        # per-segment device time (the context runs on torch's current stream)
        e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        e0.record()
        self.ctx.call(name, *dev)
        e1.record()
        self.events.append((name, e0, e1))

    events = None

    def segment_ms(self):
        """Device milliseconds per segment name, summed over the calls recorded since `events = []`."""
        self.torch.cuda.synchronize()
        out = {}
        for name, e0, e1 in self.events:
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
        return out

    def ring_gemm(self, *a, **k):
        raise NotImplementedError

    # True: the carry tree's messages stay bit strings between the parties on this GPU (the *_bits segments, what the
    # C++ engines run: they make the wire bytes on the host); False: wire bytes, one per boolean, in device memory
    msb_bits = False

    class _Msb:
        def __init__(self, be, G0, G1, P0, P1):
            self.be, self.n, self.P0, self.P1 = be, G0.size, P0, P1
            self.G0, self.G1 = G0, G1  # read in place by round 32
            self.state = be.empty(np.uint64, 6 * self.n)
            be._run("aby3_msb_begin", G0, G1, P0, P1, self.n, self.state)

        def _words(self, booleans):
            return self.be.empty(np.uint32, (booleans + 31) // 32)

        def round_pre(self, temp):
            if self.be.msb_bits and temp == 32:
                msg = self._words(self.n * 64)
                self.be._run("aby3_msb_round_pre_bits", self.state, self.n, temp, msg)
                return msg
            msg = self.be.empty(np.int8, self.n * 2 * temp)
            self.be._run("aby3_msb_round_pre", self.state, self.n, temp, msg)
            return msg

        def round_post(self, temp, msg_in):
            self.be._run("aby3_msb_round_post", self.state, self.n, temp, msg_in)

        def final(self):
            Z0, Z1 = self.be.empty(np.int8, self.n), self.be.empty(np.int8, self.n)
            self.be._run("aby3_msb_final", self.state, self.P0, self.P1, self.n, Z0, Z1)
            return Z0, Z1

        def round_mid(self, temp, msg_in):
            if self.be.msb_bits:
                msg = self._words(self.n * temp)
                self.be._run("aby3_msb_round_mid_bits", self.state, self.n, temp, msg_in, msg)
                return msg
            msg = self.be.empty(np.int8, self.n * temp)
            self.be._run("aby3_msb_round_mid", self.state, self.n, temp, msg_in, msg)
            return msg

        def last(self, msg_in):
            Z0, Z1 = self.be.empty(np.int8, self.n), self.be.empty(np.int8, self.n)
            self.be._run("aby3_msb_last_bits" if self.be.msb_bits else "aby3_msb_last", self.state, self.n, msg_in, Z0, Z1)
            return Z0, Z1

    def aby3_msb_begin(self, G0, G1, P0, P1):
        return ResidentBackend._Msb(self, G0, G1, P0, P1)
