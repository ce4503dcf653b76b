"""Development helper: copies the summaries of a tools/profile.sh run (gpurun_out/prof_<tag>/) into profiles/:
the rocprofv3 kernel stats CSV and a JSON digest of the PMC passes of the limb GEMM kernels."""
import collections
import csv
import glob
import json
import shutil
import sys

tag = sys.argv[1]
ROUND = sys.argv[2] if len(sys.argv) > 2 else "r2"
d = f"gpurun_out/prof_{tag}"
shutil.copy(glob.glob(d + "/trace/*kernel_stats.csv")[0], f"profiles/{ROUND}_bench_kernel_stats.csv")
out = {}
for p in ("pmc_rd", "pmc_wr", "pmc_sq"):
    for f in glob.glob(f"{d}/{p}/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            name = ("limb_gemm_kernel" if "limb_gemm" in k else "split_limbs_kernel" if "split_limbs" in k
                    else "rns_gemm_wide_kernel" if "rns_gemm_wide" in k else "rns_gemm_kernel" if "rns_gemm" in k else "rns_split_kernel" if "rns_split" in k
                    else "rns_combine_kernel" if "rns_combine" in k else None)
            if name:
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                agg[(name, r["Counter_Name"])].append((float(r["Counter_Value"]), dur))
        for (name, cn), v in agg.items():
            out.setdefault(name, {})[cn] = {"launches": len(v), "avg_value": sum(x for x, _ in v) / len(v),
                                            "avg_duration_us": sum(t for _, t in v) / len(v)}
for gname in ("limb_gemm_kernel", "rns_gemm_kernel", "rns_gemm_wide_kernel"):
    if gname not in out:
        continue
    g = out[gname]
    cyc = g["GRBM_GUI_ACTIVE"]["avg_value"] / 8
    g["derived"] = {
        "hbm_read_bytes_per_launch": g["FETCH_SIZE"]["avg_value"] * 1024 * 2,
        "hbm_write_bytes_per_launch": g["WRITE_SIZE"]["avg_value"] * 1024,
        "note": "FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B); "
                "separate --pmc passes of `bench.py --steps 1 --warmup 0 --no-cpu --no-bitwise --no-extras`",
        "effective_clock_GHz": cyc / g["GRBM_GUI_ACTIVE"]["avg_duration_us"] / 1e3,
        "mfma_int8_ops": g["SQ_INSTS_VALU_MFMA_MOPS_I8"]["avg_value"] * 512,
        "mfma_pipe_utilisation_at_held_clock": g["SQ_INSTS_VALU_MFMA_MOPS_I8"]["avg_value"] * 512 / (256 * 4 * 2048) / cyc,
    }
    if "SQ_VALU_MFMA_BUSY_CYCLES" in g:  # summed over the 1024 SIMDs
        g["derived"]["mfma_busy_cycles_per_simd_cycle"] = g["SQ_VALU_MFMA_BUSY_CYCLES"]["avg_value"] / (1024 * cyc)
    print(gname, json.dumps(g["derived"], indent=1))
json.dump(out, open(f"profiles/{ROUND}_pmc_summary.json", "w"), indent=1)
