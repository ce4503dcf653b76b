import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpc_amd import Context
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
ctx = Context(0); ctx.call("set_stream", torch.cuda.current_stream().cuda_stream)
r, l, c = 4096, 8192, 4096
g = torch.Generator(device=dev); g.manual_seed(1)
def rnd(n): return torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
cases = {"random": (rnd(r * l), rnd(l * c)), "zeros": (torch.zeros(r * l, dtype=torch.int64, device=dev), torch.zeros(l * c, dtype=torch.int64, device=dev)),
         "small (|x| < 2^7: one nonzero digit)": (torch.randint(-100, 100, (r * l,), dtype=torch.int64, device=dev, generator=g), torch.randint(-100, 100, (l * c,), dtype=torch.int64, device=dev, generator=g))}
cases["A zeros, B random"] = (cases["zeros"][0], cases["random"][1])
Z = torch.empty(r * c, dtype=torch.int64, device=dev)
for name, (A, B) in cases.items():
    for _ in range(3): ctx.call("ring_gemm_u64", A, B, Z, r, l, c, 0, 0, 0, 0)
    torch.cuda.synchronize()
    ctx.call("profile", 1)
    for _ in range(6): ctx.call("ring_gemm_u64", A, B, Z, r, l, c, 0, 0, 0, 0)
    torch.cuda.synchronize()
    import ctypes
    ms, cnt = ctypes.c_double(), ctypes.c_uint64()
    ctx._call("mpcx_profile_read", b"limb_gemm", ctypes.byref(ms), ctypes.byref(cnt))
    ctx.call("profile", 0)
    print(f"{name}: {ms.value / cnt.value:.3f} ms per launch ({cnt.value} launches)", flush=True)
