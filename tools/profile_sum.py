"""Development helper: summarises a tools/profile.sh output directory (kernel stats + PMC passes)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
for f in glob.glob(d + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("== kernel stats (top 14 by total time)")
    for r in rows[:14]:
        print("  %-60s calls %6s  avg_us %10.1f  total_ms %9.2f  %5.1f%%" % (
            r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
for tag in ("pmc_rd", "pmc_wr", "pmc_sq"):
    for f in glob.glob(d + "/" + tag + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if any(s in k for s in ("limb_gemm", "split_limbs", "rns_")):
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                agg[(k.split("(")[0][-40:], r["Counter_Name"])].append((float(r["Counter_Value"]), dur))
        print("==", tag)
        for k, v in sorted(agg.items()):
            n = len(v)
            print("  %-40s %-28s n=%d avg=%.6g dur_us=%.1f" % (k[0], k[1], n, sum(x for x, _ in v) / n, sum(t for _, t in v) / n))
