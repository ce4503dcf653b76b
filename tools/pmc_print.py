import collections
import csv
import glob

for f in glob.glob("gpurun_out/pmc_tmp/*counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "limb_gemm" in r["Kernel_Name"] or "pc_party" in r["Kernel_Name"] or "prg_fill" in r["Kernel_Name"]:
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append((float(r["Counter_Value"]), dur))
    for k, v in agg.items():
        n = len(v)
        print(k, "n=%d avg=%.4g dur_us=%.1f" % (n, sum(x for x, _ in v) / n, sum(d for _, d in v) / n))
