"""Does running the memory-bound passes of one ring product (residue split, CRT combine) beside the matrix-core pass
of another buy time?  Upper bound for any pipelining inside the library: N whole products of 4096 x 8192 x 4096 on ONE
context (one stream: split -> gemm -> combine strictly in turn) against the same N products dealt alternately to TWO
contexts with streams of their own, which the GPU is free to overlap in any way it can.  Random operands (what secret
shares are); wall clock around a synchronised region.  Usage: python tools/overlap_probe.py [N]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mpc_amd import Context  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ctxs = [Context(0), Context(0)]  # each with a non-blocking stream and a workspace of its own
r, l, c = 4096, 8192, 4096
g = torch.Generator(device=dev)
g.manual_seed(1)


def rnd(n):
    return torch.randint(-2 ** 63, 2 ** 63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)


A, B = [rnd(r * l), rnd(r * l)], [rnd(l * c), rnd(l * c)]
Z = [torch.empty(r * c, dtype=torch.int64, device=dev) for _ in range(2)]
torch.cuda.synchronize()


def run(two):
    for i in range(N):
        k = i % 2 if two else 0
        ctxs[k].call("ring_gemm_u64", A[k], B[k], Z[k], r, l, c, 0, 0, 0, 0)
    for cx in ctxs:
        cx.sync()


out = {}
for label, two in (("one_stream", False), ("two_streams", True), ("one_stream_again", False), ("two_streams_again", True)):
    run(two)  # warm-up: workspaces, clocks
    t0 = time.perf_counter()
    run(two)
    out[label + "_ms_per_product"] = round((time.perf_counter() - t0) * 1e3 / N, 4)
ref = [Z[0].clone(), Z[1].clone()]
run(False)
out["results_identical"] = bool(torch.equal(Z[0], ref[0]))
out["products"] = N
out["shape"] = [r, l, c]
print(json.dumps(out), flush=True)
