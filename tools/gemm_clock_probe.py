"""The clock the chip holds inside rns_gemm_wide_kernel: shader ticks / 100 MHz reference ticks, stamped once around
the K loop of every workgroup in a diagnostic build (tools/build_probes.sh -> tools/libmpcx_probes.so), read after at
least two seconds of back-to-back products of 4096 x 8192 x 4096 on each kind of operand.  Median over the workgroups
of the last launch.  Usage (the probe library must be the one the process loads):
    MPCX_LIBRARY=tools/libmpcx_probes.so python tools/gemm_clock_probe.py > profiles/r3_gemm_clock.json"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import mpc_amd  # noqa: E402
from mpc_amd import Context  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ctx = Context(0)
ctx.call("set_stream", torch.cuda.current_stream().cuda_stream)
lib = mpc_amd.load_library()
if not hasattr(lib, "mpcx_dev_rns_clock_read"):
    raise SystemExit("load the diagnostic build: MPCX_LIBRARY=tools/libmpcx_probes.so")
r, l, c = 4096, 8192, 4096
g = torch.Generator(device=dev)
g.manual_seed(1)


def rnd(n, lo=-2 ** 63, hi=2 ** 63 - 1):
    return torch.randint(lo, hi, (n,), dtype=torch.int64, device=dev, generator=g)


zeros = torch.zeros(r * l, dtype=torch.int64, device=dev)
cases = [("random", rnd(r * l), rnd(l * c)), ("zeros", zeros, zeros), ("A_zero_B_random", zeros, rnd(l * c)),
         ("small_abs_below_100", rnd(r * l, -100, 100), rnd(l * c, -100, 100)), ("random_again", rnd(r * l), rnd(l * c))]
Z = torch.empty(r * c, dtype=torch.int64, device=dev)
tiles = -(-r // 384) * (c // 256) * 18
out = {"shape": [r, l, c], "kernel": "rns_gemm_wide_kernel", "workgroups": tiles, "min_seconds_per_case": 2.0,
       "reference_clock_MHz": 100, "cases": {}}
for name, A, B in cases:
    ms, cnt = ctypes.c_double(), ctypes.c_uint64()
    t0 = time.perf_counter()
    launches = 0
    ctx.call("profile", 1)
    while time.perf_counter() - t0 < 2.0:  # back to back: the queue never drains
        for _ in range(20):
            ctx.call("ring_gemm_u64", A, B, Z, r, l, c, 0, 0, 0, 0)
        launches += 20
        torch.cuda.synchronize()
    ctx._call("mpcx_profile_read", b"limb_gemm", ctypes.byref(ms), ctypes.byref(cnt))
    ctx.call("profile", 0)
    buf = np.zeros(2 * tiles, dtype=np.uint64)
    assert lib.mpcx_dev_rns_clock_read(ctypes.c_void_p(buf.ctypes.data), ctypes.c_uint64(tiles)) == 0
    clk, ref = buf[0::2].astype(np.float64), buf[1::2].astype(np.float64)
    ok = ref > 0
    ghz = clk[ok] / ref[ok] * 0.1
    out["cases"][name] = {"launches": launches, "gemm_ms_per_launch": round(ms.value / cnt.value, 4),
                          "clock_GHz_median": round(float(np.median(ghz)), 4),
                          "clock_GHz_p10_p90": [round(float(np.percentile(ghz, 10)), 4), round(float(np.percentile(ghz, 90)), 4)],
                          "loop_us_median": round(float(np.median(ref[ok])) / 100.0, 2),
                          "loop_shader_cycles_median": int(np.median(clk[ok])), "workgroups_read": int(ok.sum())}
print(json.dumps(out, indent=1))
