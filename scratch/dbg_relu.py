import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import protocols as pr
from backends import GpuBackend, OracleBackend
from oracle_lib import prg_pops
gpu, orc = GpuBackend(), OracleBackend()
for lg in (14, 17, 20):
    n = 1 << lg
    X = prg_pops(21, 0, n)
    res = []
    for be in (gpu, orc):
        w = pr.World(be)
        Xs = pr.aby3_share(w, X, n)
        Gs, Ps = pr.aby3_add_prepare(w, Xs)
        Zb = pr.aby3_msb(w, Gs, Ps)
        C = pr.aby3_bit_inject(w, Xs, Zb)
        res.append((w, Zb, C))
    for (tag, a), (_, b) in zip(res[0][0].log, res[1][0].log):
        if not np.array_equal(a, b):
            bad = np.nonzero(a != b)[0]
            print(lg, "DIFF", tag, bad.size, bad[:5], bad[-5:])
            break
    else:
        print(lg, "logs equal")
    ok = all(np.array_equal(res[0][2][j][i], res[1][2][j][i]) for j in range(3) for i in range(2))
    print(lg, "C equal", ok)
