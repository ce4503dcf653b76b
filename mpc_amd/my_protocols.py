"""Three-party drivers of the MyEngine protocols (mpc_engine/MyEngine.cpp), in the style of protocols.py: one
thread plays all three parties, every message is logged so that two backends can be compared byte for byte.
Shares are [share of party 0, share of party 1]."""
import numpy as np

from .keys import KEY01, KEY12, KEY20, KEYS

# party 2's prgs[0] / prgs[1] (MyEngine.cpp:21-24); party j's prgs[2] is the same stream (:11-19)
SEED2 = {0: KEY20, 1: KEY12}


def my_share(w, X, n):
    """share / share_matrix (:54-89): the SecureNN sharing out of the stream parties 0 and 1 have in common."""
    return [w.be.snn_share_local(0, None, n, KEY01, w.take(0, 1, n)), w.be.snn_share_local(1, X, n, KEY01, w.take(1, 0, n))]


def my_reveal(w, S):
    return w.be.ring_add(S[0], S[1])


def my_reveal_bools(w, S):
    return w.be.xor_bools(S[0], S[1])


def my_matmul(w, Xs, Ys, r, l, c, layout=0):
    """matrix_mul (:145-219): one round."""
    be = w.be
    C0, C1 = be.my_beaver_dealer_matmul(KEY20, KEY12, KEYS, r, l, c)
    w.rec("my_matmul.P2->P0", C0)
    w.rec("my_matmul.P2->P1", C1)
    ef = [be.my_beaver_party_open_matmul(Xs[j], Ys[j], layout, SEED2[j], r, l, c) for j in range(2)]
    w.rec("my_matmul.P0->P1", ef[0])
    w.rec("my_matmul.P1->P0", ef[1])
    C = [C0, C1]
    return [be.my_beaver_party_finish_matmul(j, SEED2[j], C[j], ef[j], ef[1 - j], r, l, c, layout) for j in range(2)]


def my_vecmul(w, Xs, Ys):
    """vector_mul (:221-280): one round."""
    be = w.be
    n = Xs[0].size
    C0, C1 = be.my_beaver_dealer_vecmul(KEY20, KEY12, KEYS, n, 0, n)
    w.rec("my_vecmul.P2->P0", C0)
    w.rec("my_vecmul.P2->P1", C1)
    ef = [be.my_beaver_party_open_vecmul(Xs[j], Ys[j], SEED2[j], n, 0) for j in range(2)]
    w.rec("my_vecmul.P0->P1", ef[0])
    w.rec("my_vecmul.P1->P0", ef[1])
    C = [C0, C1]
    return [be.my_beaver_party_finish_vecmul(j, SEED2[j], C[j], ef[j], ef[1 - j], n, 0) for j in range(2)]


def my_private_compare(w, xbits, beta, r):
    """privateCompare (:287-437): beta' at party 2."""
    be = w.be
    d0 = be.my_pc_party(0, xbits[0], beta, r, 0, KEY01)
    d1 = be.my_pc_party(1, xbits[1], beta, r, 0, KEY01)
    w.rec("my_pc.P0->P2", d0)
    w.rec("my_pc.P1->P2", d1)
    b = be.snn_pc_helper(d0, d1)
    w.rec("my_pc.beta2", b)
    return b


def my_msb(w, As):
    """msb (:442-597): three rounds; returns XOR shares (one byte per element) of [a < 0]."""
    be = w.be
    n = As[0].size
    l0, xb0, l1, xb1 = be.my_msb_helper_deal(KEY20, KEY12, KEYS, 0, n)
    for tag, m in (("l0", l0), ("xb0", xb0), ("l1", l1), ("xb1", xb1)):
        w.rec("my_msb.P2." + tag, m)
    rj = [be.my_msb_party_open(As[j], SEED2[j], 0) for j in range(2)]
    w.rec("my_msb.r0", rj[0])
    w.rec("my_msb.r1", rj[1])
    beta0, R = be.my_msb_party_pc_inputs(rj[0], rj[1], 0, KEY01)
    if w.record:  # party 1 derives the same pair
        b2, R2 = be.my_msb_party_pc_inputs(rj[1], rj[0], 0, KEY01)
        assert np.array_equal(beta0, b2) and np.array_equal(R, R2)
    beta1 = my_private_compare(w, [xb0, xb1], beta0, R)
    s0, s1 = be.my_msb_helper_beta(beta1, n, 0, KEYS)
    w.rec("my_msb.P2.b0", s0)
    w.rec("my_msb.P2.b1", s1)
    return [be.my_msb_party_finish(0, s0, beta0, l0, R), be.my_msb_party_finish(1, s1, beta0, l1, R)]


def my_bit_inject(w, Xs, As):
    """bit_inject (:599-772): one round; arithmetic shares of x * a for a boolean-shared a."""
    be = w.be
    n = Xs[0].size
    D = be.my_bit_inject_dealer(KEY20, KEY12, KEYS, n, 0, n)
    for tag, m in zip(("V0", "W0", "S0", "T0", "V1", "W1", "S1", "T1"), D):
        w.rec("my_bit_inject.P2." + tag, m)
    pre = [be.my_bit_inject_party_pre(Xs[j], As[j], SEED2[j], 0) for j in range(2)]
    for j in range(2):
        w.rec(f"my_bit_inject.y{j}", pre[j][0])
        w.rec(f"my_bit_inject.c{j}", pre[j][1])
    return [be.my_bit_inject_party_post(j, SEED2[j], 0, pre[j][0], pre[j][1], pre[1 - j][0], pre[1 - j][1],
                                        *D[4 * j: 4 * j + 4]) for j in range(2)]
