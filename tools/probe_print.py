import json
ok = False
for l in open("gpurun_out/probe.log"):
    if l.startswith("{"):
        d = json.loads(l)
        ok = True
        print("ms/step", round(d["ms_per_step"], 2), "gemm ms", round(d["roofline"]["avg_launch_ms"], 3), "frac",
              round(d["roofline"]["frac"], 3), d["kernel_ms_per_step"])
if not ok:
    print(open("gpurun_out/probe.log").read()[-1500:])
