"""Element-range runs of the protocol drivers: a vector is cut into contiguous ranges (one per GPU), each range
is run as its own `World(n_total, i0)`, and the per-range messages and outputs are put back together.  The result
must equal the whole-vector run message by message."""
import numpy as np

import protocols as pr

# messages made of sections of n entries each (the ranges of one section concatenate): ABY3 bit_inject
# (ABY3Engine.cpp:600-606 and the like).  Every other message is element-major.
SECTIONED = ("bit_inject.",)


def join_logs(logs, counts):
    """Per-range message logs -> the whole vector's log."""
    out = []
    for entries in zip(*logs):
        tag = entries[0][0]
        assert all(t == tag for t, _ in entries)
        arrs = [a for _, a in entries]
        if tag.startswith(SECTIONED):
            k = arrs[0].size // counts[0]
            arrs = [a.reshape(k, m) for a, m in zip(arrs, counts)]
            out.append((tag, np.concatenate(arrs, axis=1).ravel()))
        else:
            out.append((tag, np.concatenate(arrs)))
    return out


def same_logs(got, want):
    assert [t for t, _ in got] == [t for t, _ in want]
    for (tag, a), (_, b) in zip(got, want):
        assert a.dtype == b.dtype and np.array_equal(a, b), f"message {tag} differs"


def aby3_relu(w, x):
    """test/test_ABY3.cpp:196-214: share -> add_prepare -> msb -> bit_inject; returns, per party, both shares of x * msb(x)
    followed by both shares of x - x * msb(x)."""
    Xs = pr.aby3_share(w, x, x.size)
    Gs, Ps = pr.aby3_add_prepare(w, Xs)
    Zb = pr.aby3_msb(w, Gs, Ps)
    pops = [list(p) for p in w.pops]
    C = pr.aby3_bit_inject(w, Xs, Zb)
    # and once more with the minuend (x - x * msb(x), the post segment and the subtraction in one call where the
    # backend has it), from the same stream positions and without logging the messages a second time
    w2 = pr.World(w.be, record=False, n_total=w.n_total, i0=w.i0)
    w2.pops = pops
    R = pr.aby3_bit_inject(w2, Xs, Zb, minuend=Xs)
    return [np.concatenate([np.asarray(c[0]), np.asarray(c[1]), np.asarray(r[0]), np.asarray(r[1])]) for c, r in zip(C, R)]


def snn_msb(w, a):
    """test/test_securenn.cpp:283-308: share -> msb; returns the two parties' shares of the sign bit."""
    As = pr.snn_share(w, a, a.size)
    B = pr.snn_msb(w, As[:2])
    return [np.asarray(B[0]), np.asarray(B[1])]


def run_in_ranges(be, fn, x, ranges, halves=1):
    """fn(world, slice) per range; returns (joined outputs, joined log).  halves = arrays packed in one output."""
    n = x.size
    assert ranges[0][0] == 0 and sum(m for _, m in ranges) == n
    outs, logs = [], []
    for i0, m in ranges:
        w = pr.World(be, n_total=n, i0=i0)
        outs.append(fn(w, x[i0:i0 + m]))
        logs.append(w.log)
    counts = [m for _, m in ranges]
    joined = []
    for per_party in zip(*outs):
        parts = [o.reshape(halves, m) for o, m in zip(per_party, counts)]
        joined.append(np.concatenate(parts, axis=1).ravel())
    return joined, join_logs(logs, counts)
