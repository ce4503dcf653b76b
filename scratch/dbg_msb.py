import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mpc_amd import Context
ctx = Context(0)
n = 1 << 20
rng = np.random.default_rng(5)
G0, G1, P0, P1 = [rng.integers(0, 2**64, n, dtype=np.uint64) for _ in range(4)]
EM = {1: 0x5555555555555555, 2: 0x1111111111111111, 4: 0x0101010101010101, 8: 0x0001000100010001, 16: 0x0000000100000001, 32: 1}
def model_begin(G0, G1, P0, P1):
    top = np.uint64(1 << 63)
    return [G0 & ~top, G1 & ~top, P0 | top, P1 | top]
def products(st, s):
    E = np.uint64(EM[s]); sh = np.uint64(s)
    g0, g1, p0, p1 = st
    po0, po1 = (p0 >> sh) & E, (p1 >> sh) & E
    ge0, ge1, pe0, pe1 = g0 & E, g1 & E, p0 & E, p1 & E
    re = (po0 & ge0) ^ (po1 & ge0) ^ (po0 & ge1)
    ro = (po0 & pe0) ^ (po1 & pe0) ^ (po0 & pe1)
    return re, ro
for rep in range(2):
    st_d = ctx.empty(np.uint64, 4 * n)
    ctx.call("aby3_msb_begin", ctx.to_device(G0), ctx.to_device(G1), ctx.to_device(P0), ctx.to_device(P1), n, st_d)
    ctx.sync()
    st = model_begin(G0, G1, P0, P1)
    got = st_d.to_host().reshape(4, n)
    print("begin ok", all(np.array_equal(got[i], st[i]) for i in range(4)))
    temp = 32
    while temp >= 1:
        s = 32 // temp
        msg = ctx.empty(np.int8, 2 * temp * n)
        ctx.call("aby3_msb_round_pre", st_d, n, temp, msg)
        ctx.sync()
        re, ro = products(st, s)
        E = np.uint64(EM[s]); sh = np.uint64(s)
        want0 = re ^ ((st[0] >> sh) & E); want2 = ro
        got = st_d.to_host().reshape(4, n)
        bad0 = np.nonzero(got[0] != want0)[0]; bad2 = np.nonzero(got[2] != want2)[0]
        print(rep, temp, "pre: share0 G bad", bad0.size, bad0[:4], "P bad", bad2.size, bad2[:4],
              "untouched ok", np.array_equal(got[1], st[1]) and np.array_equal(got[3], st[3]))
        m = msg.to_host().reshape(n, temp, 2)
        pos = (2 * np.arange(temp) * s).astype(np.uint64)
        wre = ((re[:, None] >> pos[None, :]) & np.uint64(1)).astype(np.int8)
        wro = ((ro[:, None] >> pos[None, :]) & np.uint64(1)).astype(np.int8)
        print("   msg even bad", int((m[:, :, 0] != wre).sum()), "odd bad", int((m[:, :, 1] != wro).sum()))
        # post with own message (any message works for the check)
        ctx.call("aby3_msb_round_post", st_d, n, temp, msg)
        ctx.sync()
        want1 = re ^ ((st[1] >> sh) & E); want3 = ro
        got = st_d.to_host().reshape(4, n)
        bad1 = np.nonzero(got[1] != want1)[0]; bad3 = np.nonzero(got[3] != want3)[0]
        print(rep, temp, "post: share1 G bad", bad1.size, bad1[:4], "P bad", bad3.size, bad3[:4])
        st = [want0, want1, want2, want3]
        ctx.call("h2d", st_d, 0, 0) if False else None
        # resync device state to the model so later rounds are checked independently
        st_d = ctx.to_device(np.concatenate(st))
        temp //= 2
