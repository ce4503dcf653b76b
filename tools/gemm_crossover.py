"""Development helper: times mpcx_ring_gemm_u64 in the digit-pair (MPCX_GEMM_RNS=0) and the residue (=2) formulation
on a list of shapes, to place the automatic threshold (ring_gemm_use_rns)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpc_amd import Context
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
ctx = Context(0); ctx.call("set_stream", torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev); g.manual_seed(1)
def rnd(n): return torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
shapes = [(1024, 2048, 1024), (1536, 3072, 1536), (2048, 2048, 2048), (2048, 4096, 2048), (2560, 2048, 2560), (2560, 5120, 2560),
          (3072, 6144, 3072), (4096, 1024, 4096), (4096, 8192, 4096), (128, 8192, 8192), (512, 4096, 8192)]
for r, l, c in shapes:
    A, B, Z = rnd(r * l), rnd(l * c), torch.empty(r * c, dtype=torch.int64, device=dev)
    res = []
    for mode in ("0", "2"):
        os.environ["MPCX_GEMM_RNS"] = mode
        for _ in range(2): ctx.call("ring_gemm_u64", A, B, Z, r, l, c, 0, 0, 0, 0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): ctx.call("ring_gemm_u64", A, B, Z, r, l, c, 0, 0, 0, 0)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 5 * 1e3)
    print(f"{r}x{l}x{c}: digit-pair {res[0]:.3f} ms, residue {res[1]:.3f} ms", flush=True)
