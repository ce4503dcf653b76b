"""Three-party protocol drivers composed from per-party local segments (the methods of mpc_amd.backends.GpuBackend,
or of any object with the same method names).

Each driver plays all three parties in one thread: it runs every party's "before the round" segment,
moves the messages the way the reference's messenger round would (VirtuleMessenger::send_and_recv),
then runs the "after the round" segments.  All messages are recorded in `World.log` so that two
backends can be compared message by message.  The call order and the per-party pop counters follow
the reference engines (file:line cited per function)."""
import numpy as np

from .keys import KEY01, KEY12, KEY20, KEYS


def pair_seed(a, b):
    return {frozenset((0, 1)): KEY01, frozenset((1, 2)): KEY12, frozenset((0, 2)): KEY20}[frozenset((a, b))]


class World:
    """Per-party PRG pop counters (every party owns its own PRgenerator objects, ABY3Engine.cpp:34-53)."""

    def __init__(self, backend, record=True, n_total=None, i0=0):
        self.be = backend
        self.record = record  # False: messages stay wherever the backend keeps them (device-resident timing runs)
        # Element-range mode (one GPU's share of a vector): every vector handed to a driver holds elements
        # [i0, i0 + n) of a vector of n_total elements; PRG offsets are those of the whole vector.
        self.n_total, self.i0 = n_total, i0
        # pops[j][k]: words party j has popped from prgs[k] (k == j: the self stream)
        self.pops = [[0, 0, 0] for _ in range(3)]
        self.log = []

    def seed(self, j, k):
        return KEYS if j == k else pair_seed(j, k)

    def take(self, j, k, count):
        p = self.pops[j][k]
        self.pops[j][k] += count
        return p

    def whole(self, n):
        """Size of the whole vector of which the caller holds n elements."""
        return self.n_total if self.n_total is not None else n

    def rec(self, tag, arr):
        if self.record:
            self.log.append((tag, np.array(arr, copy=True)))


# ------------------------------------------------------------------ ABY3
def aby3_share(w, X, n):
    """ABY3Engine::share / share_matrix (ABY3Engine.cpp:63-106).  Returns shares[j] = (X_j, X_{j+1})."""
    R = []
    for j in range(3):
        nx, pv = (j + 1) % 3, (j + 2) % 3
        nt = w.whole(n)
        R.append(w.be.aby3_share_local(X if j == 0 else None, n, w.seed(j, nx), w.take(j, nx, nt) + w.i0, w.seed(j, pv),
                                       w.take(j, pv, nt) + w.i0))
        w.rec(f"share.P{j}->P{pv}", R[j])
    return [(R[j], R[(j + 1) % 3]) for j in range(3)]


def aby3_share_bools(w, X, n):
    R = []
    for j in range(3):
        nx, pv = (j + 1) % 3, (j + 2) % 3
        R.append(w.be.aby3_share_bools_local(X if j == 0 else None, n, w.seed(j, nx), w.take(j, nx, n), w.seed(j, pv),
                                             w.take(j, pv, n)))
    return [(R[j], R[(j + 1) % 3]) for j in range(3)]


def aby3_reveal(w, S):
    """reveal / reveal_matrix (ABY3Engine.cpp:132-158): party j receives X_{j-1} from prev."""
    return [w.be.aby3_reveal3(S[(j + 2) % 3][0], S[j][0], S[j][1]) for j in range(3)]


def aby3_reveal_bools(w, S):
    return [w.be.aby3_reveal3_bools(S[(j + 2) % 3][0], S[j][0], S[j][1]) for j in range(3)]


def aby3_matmul(w, Xs, Ys, r, l, c, layout=0):
    """matrix_mul (ABY3Engine.cpp:177-201).  Z shares are row-major."""
    Z0 = []
    for j in range(3):
        nx, pv = (j + 1) % 3, (j + 2) % 3
        Z0.append(w.be.aby3_matmul_local(Xs[j][0], Xs[j][1], Ys[j][0], Ys[j][1], r, l, c, layout, w.seed(j, nx),
                                         w.take(j, nx, r * c), w.seed(j, pv), w.take(j, pv, r * c)))
        w.rec(f"matmul.P{j}->P{pv}", Z0[j])
    return [(Z0[j], Z0[(j + 1) % 3]) for j in range(3)]


def aby3_vecmul(w, Xs, Ys, n):
    Z0 = []
    for j in range(3):
        nx, pv = (j + 1) % 3, (j + 2) % 3
        nt = w.whole(n)
        Z0.append(w.be.aby3_vecmul_local(Xs[j][0], Xs[j][1], Ys[j][0], Ys[j][1], n, w.seed(j, nx),
                                         w.take(j, nx, nt) + w.i0, w.seed(j, pv), w.take(j, pv, nt) + w.i0))
    return [(Z0[j], Z0[(j + 1) % 3]) for j in range(3)]


def aby3_add_prepare(w, Xs):
    """add_prepare (ABY3Engine.cpp:295-326): two rounds, each party sends to prev."""
    C0 = [w.be.aby3_add_prepare_step1(Xs[j][0], Xs[j][1]) for j in range(3)]
    for j in range(3):
        w.rec(f"add_prepare.C.P{j}", C0[j])
    out = [w.be.aby3_add_prepare_step2(Xs[j][0], Xs[j][1], C0[j], C0[(j + 1) % 3]) for j in range(3)]
    P0 = [o[0] for o in out]
    P1 = [o[1] for o in out]
    G0 = [o[2] for o in out]
    for j in range(3):
        w.rec(f"add_prepare.G.P{j}", G0[j])
    Gs = [(G0[j], G0[(j + 1) % 3]) for j in range(3)]
    Ps = [(P0[j], P1[j]) for j in range(3)]
    return Gs, Ps


MSB_FUSED = True  # drive backends that have them through round_mid / last (one call per messenger gap)


def aby3_msb(w, Gs, Ps):
    """msb (ABY3Engine.cpp:361-467): six rounds temp = 32..1, message 2*temp bytes per element to prev."""
    hs = [w.be.aby3_msb_begin(Gs[j][0], Gs[j][1], Ps[j][0], Ps[j][1]) for j in range(3)]
    if MSB_FUSED and hasattr(hs[0], "round_mid"):
        msgs = [hs[j].round_pre(32) for j in range(3)]
        temp = 32
        while True:
            for j in range(3):
                w.rec(f"msb.t{temp}.P{j}", msgs[j])
            if temp == 1:
                return [hs[j].last(msgs[(j + 1) % 3]) for j in range(3)]
            msgs = [hs[j].round_mid(temp, msgs[(j + 1) % 3]) for j in range(3)]
            temp //= 2
    temp = 32
    while temp:
        msgs = [hs[j].round_pre(temp) for j in range(3)]
        for j in range(3):
            w.rec(f"msb.t{temp}.P{j}", msgs[j])
        for j in range(3):
            hs[j].round_post(temp, msgs[(j + 1) % 3])
        temp //= 2
    return [hs[j].final() for j in range(3)]


def aby3_bit_inject(w, As, Bs, minuend=None):
    """bit_inject (ABY3Engine.cpp:510-1038).  With `minuend` (share pairs S_j) the result is S_j - C_j, the ReLU
    composition x - bit_inject(x, msb) (benchmark/benchmark1.cpp:71-73 and the subtraction): one call per party after
    the round on backends whose post segment takes the minuend, post followed by ring_sub on the others."""
    be = w.be
    n = As[0][0].size
    nt = w.whole(n)
    rng = (nt, w.i0)
    p01_0, p20_0 = w.take(0, 1, 6 * nt), w.take(0, 2, 3 * nt)
    p01_1, p12_1 = w.take(1, 0, 6 * nt), w.take(1, 2, 3 * nt)
    p20_2, p12_2 = w.take(2, 0, 3 * nt), w.take(2, 1, 3 * nt)
    m01, m02 = be.aby3_bit_inject_p0_pre(As[0][0], Bs[0][0], Bs[0][1], KEY01, p01_0, KEY20, p20_0, rng=rng)
    m10, m12 = be.aby3_bit_inject_p1_pre(As[1][0], As[1][1], Bs[1][0], Bs[1][1], KEY01, p01_1, KEY12, p12_1, rng=rng)
    m20, m21 = be.aby3_bit_inject_p2_pre(Bs[2][0], Bs[2][1], KEY20, p20_2, KEY12, p12_2, rng=rng)
    for tag, m in (("01", m01), ("02", m02), ("10", m10), ("12", m12), ("20", m20), ("21", m21)):
        w.rec("bit_inject.P%s->P%s" % (tag[0], tag[1]), m)
    fused = minuend is not None and getattr(be, "bit_inject_post_takes_minuend", False)
    kw = [dict(rng=rng, sub=minuend[j]) if fused else dict(rng=rng) for j in range(3)]
    C = [be.aby3_bit_inject_p0_post(Bs[0][0], KEY01, p01_0, KEY20, p20_0, m10, m20, **kw[0]),
         be.aby3_bit_inject_p1_post(Bs[1][1], KEY01, p01_1, KEY12, p12_1, m01, m21, **kw[1]),
         be.aby3_bit_inject_p2_post(Bs[2][0], Bs[2][1], KEY20, p20_2, KEY12, p12_2, m02, m12, **kw[2])]
    if minuend is not None and not fused:
        C = [[be.ring_sub(minuend[j][k], C[j][k]) for k in range(2)] for j in range(3)]
    return C


def aby3_trunc(w, Xs, number):
    """trunc (ABY3Engine.cpp:469-508): P1 and P2 pop the stream they share."""
    n = Xs[0][0].size
    Y0 = [w.be.aby3_trunc_local(0, Xs[0][0], Xs[0][1], number, KEY12, 0),
          w.be.aby3_trunc_local(1, Xs[1][0], Xs[1][1], number, KEY12, w.take(1, 2, w.whole(n)) + w.i0),
          w.be.aby3_trunc_local(2, Xs[2][0], Xs[2][1], number, KEY12, w.take(2, 1, w.whole(n)) + w.i0)]
    return [(Y0[j], Y0[(j + 1) % 3]) for j in range(3)]


# ------------------------------------------------------------------ SecureNN
def snn_share(w, X, n):
    """SecurennEngine::share (SecurennEngine.cpp:318-334).  Returns [X_0, X_1, zeros]."""
    nt = w.whole(n)
    return [w.be.snn_share_local(0, None, n, KEY01, w.take(0, 1, nt) + w.i0),
            w.be.snn_share_local(1, X, n, KEY01, w.take(1, 0, nt) + w.i0),
            w.be.snn_share_local(2, None, n, KEY01, 0)]


def snn_reveal(w, S):
    return w.be.ring_add(S[0], S[1])


def snn_matmul(w, Xs, Ys, r, l, c, layout=0):
    """matrix_mul (SecurennEngine.cpp:61-169)."""
    be = w.be
    t0, t1 = be.snn_beaver_dealer_matmul(KEYS, r, l, c)
    w.rec("snn_matmul.P2->P0", t0)
    w.rec("snn_matmul.P2->P1", t1)
    ef0 = be.snn_beaver_party_open_matmul(Xs[0], Ys[0], layout, t0, r, l, c)
    ef1 = be.snn_beaver_party_open_matmul(Xs[1], Ys[1], layout, t1, r, l, c)
    w.rec("snn_matmul.P0->P1", ef0)
    w.rec("snn_matmul.P1->P0", ef1)
    Z0 = be.snn_beaver_party_finish_matmul(0, t0, ef0, ef1, r, l, c, layout)
    Z1 = be.snn_beaver_party_finish_matmul(1, t1, ef1, ef0, r, l, c, layout)
    return [Z0, Z1, np.zeros(r * c, dtype=np.int64)]


def snn_vecmul(w, Xs, Ys):
    """vector_mul (SecurennEngine.cpp:171-316)."""
    be = w.be
    n = Xs[0].size
    A0, B0, C0, A1, B1, C1 = be.snn_beaver_dealer_vecmul(KEYS, w.whole(n), w.i0, n)
    for tag, m in (("A0", A0), ("B0", B0), ("C0", C0), ("A1", A1), ("B1", B1), ("C1", C1)):
        w.rec("snn_vecmul.P2." + tag, m)
    E0, F0 = be.snn_beaver_party_open_vecmul(Xs[0], Ys[0], A0, B0)
    E1, F1 = be.snn_beaver_party_open_vecmul(Xs[1], Ys[1], A1, B1)
    for tag, m in (("E0", E0), ("F0", F0), ("E1", E1), ("F1", F1)):
        w.rec("snn_vecmul." + tag, m)
    Z0 = be.snn_beaver_party_finish_vecmul(0, A0, B0, C0, E0, F0, E1, F1)
    Z1 = be.snn_beaver_party_finish_vecmul(1, A1, B1, C1, E1, F1, E0, F0)
    return [Z0, Z1, None]


def snn_private_compare(w, xbits, beta, r):
    """privateCompare (SecurennEngine.cpp:507-680).  xbits = [share of P0, share of P1]; returns beta' at P2."""
    be = w.be
    n = beta.size
    nt, i0 = w.whole(n), w.i0
    d0 = be.snn_pc_party(0, xbits[0], beta, r, nt, i0, KEY01)
    d1 = be.snn_pc_party(1, xbits[1], beta, r, nt, i0, KEY01)
    w.rec("pc.P0->P2", d0)
    w.rec("pc.P1->P2", d1)
    b = be.snn_pc_helper(d0, d1)
    w.rec("pc.beta2", b)
    return b


def snn_share_convert(w, As):
    """shareConvert (SecurennEngine.cpp:685-893).  As = [a_0, a_1]; returns [y_0, y_1] over Z_{2^64-1}."""
    be = w.be
    n = As[0].size
    nt, i0 = w.whole(n), w.i0
    x0 = be.snn_sc_party_open(0, As[0], nt, i0, KEY01)
    x1 = be.snn_sc_party_open(1, As[1], nt, i0, KEY01)
    w.rec("sc.x0", x0)
    w.rec("sc.x1", x1)
    xb0, dl0, xb1, dl1 = be.snn_sc_helper_split(x0, x1, nt, i0, KEYS)
    for tag, m in (("xb0", xb0), ("dl0", dl0), ("xb1", xb1), ("dl1", dl1)):
        w.rec("sc.P2." + tag, m)
    eta2, rm1 = be.snn_sc_party_pc_inputs(nt, i0, n, KEY01)
    eta1 = snn_private_compare(w, [xb0, xb1], eta2, rm1)
    e0, e1 = be.snn_sc_helper_eta(eta1, nt, i0, KEYS)
    w.rec("sc.P2.eta0", e0)
    w.rec("sc.P2.eta1", e1)
    y0 = be.snn_sc_party_finish(0, As[0], dl0, e0, nt, i0, KEY01)
    y1 = be.snn_sc_party_finish(1, As[1], dl1, e1, nt, i0, KEY01)
    return [y0, y1]


def snn_msb_L1(w, Ys):
    """msb_L_1 (SecurennEngine.cpp:898-1084).  Ys over Z_{2^64-1}; returns shares of the msb over Z_2^64."""
    be = w.be
    n = Ys[0].size
    nt, i0 = w.whole(n), w.i0
    x0, l0, xb0, x1, l1, xb1 = be.snn_msbl1_helper_deal(nt, i0, n, KEYS)
    for tag, m in (("x0", x0), ("l0", l0), ("xb0", xb0), ("x1", x1), ("l1", l1), ("xb1", xb1)):
        w.rec("msbl1.P2." + tag, m)
    r0 = be.snn_msbl1_party_open(Ys[0], x0)
    r1 = be.snn_msbl1_party_open(Ys[1], x1)
    w.rec("msbl1.r0", r0)
    w.rec("msbl1.r1", r1)
    beta0, R = be.snn_msbl1_party_pc_inputs(r0, r1, i0, KEY01)
    if w.record:  # party 1 derives the same pair
        beta0b, Rb = be.snn_msbl1_party_pc_inputs(r1, r0, i0, KEY01)
        assert np.array_equal(R, Rb) and np.array_equal(beta0, beta0b)
    beta1 = snn_private_compare(w, [xb0, xb1], beta0, R)
    b0, b1 = be.snn_msbl1_helper_beta(beta1, nt, i0, KEYS)
    w.rec("msbl1.P2.b0", b0)
    w.rec("msbl1.P2.b1", b1)
    g0, d0 = be.snn_msbl1_party_mid(0, b0, l0, beta0, R)
    g1, d1 = be.snn_msbl1_party_mid(1, b1, l1, beta0, R)
    th = snn_vecmul(w, [g0, g1], [d0, d1])
    B0 = be.snn_msbl1_party_finish(g0, d0, th[0])
    B1 = be.snn_msbl1_party_finish(g1, d1, th[1])
    return [B0, B1]


SNN_FUSED = True  # drive backends that have them through the one-call-per-gap msb segments


def snn_msb(w, As):
    """msb (SecurennEngine.cpp:1089-1101): shareConvert(2a) then msb_L_1."""
    if SNN_FUSED and hasattr(w.be, "snn_msb_party_open"):
        return snn_msb_fused(w, As)
    twoA = [w.be.ring_shl1(As[0]), w.be.ring_shl1(As[1])]
    return snn_msb_L1(w, snn_share_convert(w, twoA))


def snn_msb_fused(w, As):
    """The same rounds and messages as snn_msb, with the local work of parties 0 / 1 done as one call per gap between
    two messenger rounds (the message tags are those of the separate segments, so the logs compare one to one)."""
    be = w.be
    n = As[0].size
    nt, i0 = w.whole(n), w.i0
    # shareConvert (SecurennEngine.cpp:685-893) on 2a
    x0 = be.snn_msb_party_open(0, As[0], nt, i0, KEY01)
    x1 = be.snn_msb_party_open(1, As[1], nt, i0, KEY01)
    w.rec("sc.x0", x0)
    w.rec("sc.x1", x1)
    xb0, dl0, xb1, dl1 = be.snn_sc_helper_split(x0, x1, nt, i0, KEYS)
    for tag, m in (("xb0", xb0), ("dl0", dl0), ("xb1", xb1), ("dl1", dl1)):
        w.rec("sc.P2." + tag, m)
    eta2, rm1 = be.snn_sc_party_pc_inputs(nt, i0, n, KEY01)
    eta1 = snn_private_compare(w, [xb0, xb1], eta2, rm1)
    e0, e1 = be.snn_sc_helper_eta(eta1, nt, i0, KEYS)
    w.rec("sc.P2.eta0", e0)
    w.rec("sc.P2.eta1", e1)
    # msb_L_1 (:898-1084)
    X0, l0, xbb0, X1, l1, xbb1 = be.snn_msbl1_helper_deal(nt, i0, n, KEYS)
    for tag, m in (("x0", X0), ("l0", l0), ("xb0", xbb0), ("x1", X1), ("l1", l1), ("xb1", xbb1)):
        w.rec("msbl1.P2." + tag, m)
    r0 = be.snn_msb_party_convert_open(0, As[0], dl0, e0, X0, nt, i0, KEY01)
    r1 = be.snn_msb_party_convert_open(1, As[1], dl1, e1, X1, nt, i0, KEY01)
    w.rec("msbl1.r0", r0)
    w.rec("msbl1.r1", r1)
    beta0, R = be.snn_msb_party_pc_inputs(r0, r1, i0, KEY01)
    if w.record:  # party 1 derives the same pair
        beta0b, Rb = be.snn_msb_party_pc_inputs(r1, r0, i0, KEY01)
        assert np.array_equal(R, Rb) and np.array_equal(beta0, beta0b)
    beta1 = snn_private_compare(w, [xbb0, xbb1], beta0, R)
    b0, b1 = be.snn_msbl1_helper_beta(beta1, nt, i0, KEYS)
    w.rec("msbl1.P2.b0", b0)
    w.rec("msbl1.P2.b1", b1)
    A0, B0, C0, A1, B1, C1 = be.snn_beaver_dealer_vecmul(KEYS, nt, i0, n)
    for tag, m in (("A0", A0), ("B0", B0), ("C0", C0), ("A1", A1), ("B1", B1), ("C1", C1)):
        w.rec("snn_vecmul.P2." + tag, m)
    g0, d0, E0, F0 = be.snn_msb_party_mid_open(0, b0, l0, beta0, R, A0, B0)
    g1, d1, E1, F1 = be.snn_msb_party_mid_open(1, b1, l1, beta0, R, A1, B1)
    for tag, m in (("E0", E0), ("F0", F0), ("E1", E1), ("F1", F1)):
        w.rec("snn_vecmul." + tag, m)
    return [be.snn_msb_party_finish(0, A0, B0, C0, E0, F0, E1, F1, g0, d0),
            be.snn_msb_party_finish(1, A1, B1, C1, E1, F1, E0, F0, g1, d1)]
