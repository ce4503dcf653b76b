"""Development helper: time of mpcx_ring_gemm_u64 on square shapes (run once per MPCX_GEMM_RNS setting)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpc_amd import Context

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ctx = Context(0)
ctx.call("set_stream", torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev)
g.manual_seed(1)
for n in [int(v) for v in os.environ.get("PROBE_SIZES", "128,256,512,768,1024,1536,2048,4096").split(",")]:
    A = torch.randint(-2 ** 63, 2 ** 63 - 1, (n * n,), dtype=torch.int64, device=dev, generator=g)
    B = torch.randint(-2 ** 63, 2 ** 63 - 1, (n * n,), dtype=torch.int64, device=dev, generator=g)
    Z = torch.empty(n * n, dtype=torch.int64, device=dev)
    for _ in range(3):
        ctx.call("ring_gemm_u64", A, B, Z, n, n, n, 0, 0, 0, 0)
    torch.cuda.synchronize()
    reps = 20 if n <= 1024 else 6
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.call("ring_gemm_u64", A, B, Z, n, n, n, 0, 0, 0, 0)
    torch.cuda.synchronize()
    print(f"n={n}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms", flush=True)
