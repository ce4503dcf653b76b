"""Development helper: digest of tools/pmc_bitwise.sh -- per kernel, the average HBM bytes per launch (FETCH_SIZE
doubled as gfx950 requires, WRITE_SIZE; both in KiB) beside the algorithmic bytes of a 2^24-element launch."""
import collections
import csv
import glob
import json

N = 1 << 24
# algorithmic bytes per element and launch
ALGO = {
    "pc_party_kernel": ("265 B protocol data + 64 B cached s bytes (+ 64 B of u for beta = 1, r = -1)", 329),
    "pc_party_gen_kernel": ("265 B protocol data + 48 B of checkpoints (+ 64 B of u for beta = 1, r = -1)", 313),
    "pc_helper_kernel": ("2 x 128 B in + 1 B out", 257),
    "bit_share_kernel": ("8 B + 64 B random bytes in, 2 x 128 B out", 328),
    "bit_share_gen_kernel": ("8 B + 48 B of checkpoints in, 2 x 128 B out", 312),
    "prg_mod67_kernel": ("64 bytes written per element and slice", 64),
    "bi_sender_pre_kernel": ("8 (or 16) + 2 B in, 40 B out, 5 x 0.375 B/word of checkpoints", 54),
    "bi_helper_pre_kernel": ("2 B in, 16 B out", 20),
    "msb_last_kernel": ("2 B message + 2 B backup + bits in, 2 B out", 6.4),
    "bi_post_kernel": ("with the ReLU minuend, averaged over the three parties: messages and select bytes 25 / 25 / 50 B "
                       "+ 16 B minuend in, 16 B out (57, 57, 82 B)", 196 / 3),
    "beaver_deal_vec_kernel": ("6 x 8 B out, 12 stream words x 0.375 B of checkpoints in", 52.5),
    "my_bi_deal_kernel": ("8 x 8 B out, 12 stream words x 0.375 B of checkpoints in", 68.5),
}
MSB = ["msb_round_pre_kernel", "msb_round_mid_kernel", "msb_round_post_kernel"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for kind, cname in (("rd", "FETCH_SIZE"), ("wr", "WRITE_SIZE")):
    for f in glob.glob(f"gpurun_out/pmc_bitwise/{kind}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != cname:
                continue
            k = r["Kernel_Name"]
            name = next((a for a in sorted(ALGO, key=len, reverse=True) + MSB if a in k), None)
            if name is None:
                continue
            if name.startswith("msb_round"):
                name += "<first level>" if ("true, true" in k or "false, true" in k) else "<later levels>"
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            agg[name][cname].append((float(r["Counter_Value"]), dur))
# third pass: VALU / LDS / wave counters of the same kernels (one launch = one row per counter)
SQ = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES",
      "SQ_INSTS_LDS", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"]
sq = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_bitwise/sq/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] not in SQ:
            continue
        k = r["Kernel_Name"]
        name = next((a for a in sorted(ALGO, key=len, reverse=True) + MSB if a in k), None)
        if name is None:
            continue
        if name.startswith("msb_round"):
            name += "<first level>" if ("true, true" in k or "false, true" in k) else "<later levels>"
        sq[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for name, c in sorted(agg.items()):
    rd = [v for v, _ in c.get("FETCH_SIZE", [])]
    wr = [v for v, _ in c.get("WRITE_SIZE", [])]
    durs = [d for _, d in c.get("FETCH_SIZE", [])]
    # launches of other sizes (warm-up of small tables etc.) are dropped: keep the 2^24-element ones = the longest half
    e = {"launches": len(rd), "avg_duration_us": sum(durs) / max(len(durs), 1),
         "hbm_read_bytes_per_launch": 2 * 1024 * sum(rd) / max(len(rd), 1),
         "hbm_write_bytes_per_launch": 1024 * sum(wr) / max(len(wr), 1)}
    base = name.split("<")[0]
    if base in ALGO:
        e["algorithmic"] = ALGO[base][0]
        e["algorithmic_bytes_per_launch"] = ALGO[base][1] * N
    if name in sq:
        # launches of other sizes are few; the per-launch averages below are over all launches of the kernel in the run
        a = {cn: sum(v) / len(v) for cn, v in sq[name].items() if v}
        e["sq_per_launch"] = {cn: round(v) for cn, v in a.items()}
        if a.get("SQ_WAVES") and a.get("GRBM_GUI_ACTIVE"):
            e["valu_insts_per_wave"] = round(a["SQ_INSTS_VALU"] / a["SQ_WAVES"], 1)
            # GRBM_GUI_ACTIVE arrives summed over the 8 XCDs; SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU on this stack (no
            # busy-cycle figure), so the rate is given in wave instructions per SIMD cycle; what a SIMD sustains per
            # instruction kind is in profiles/r3_valu_rate_probe.txt (tools/valu_rate_probe.hip)
            cycles = a["GRBM_GUI_ACTIVE"] / 8
            e["clock_GHz"] = round(cycles / (e["avg_duration_us"] * 1e3), 3)
            e["valu_wave_insts_per_simd_cycle"] = round(a["SQ_INSTS_VALU"] / (cycles * 1024), 4)
            e["waves_per_simd_avg"] = round(a["SQ_WAVE_CYCLES"] * 4 / (cycles * 1024), 2)
            e["wave_cycles_issue_stalled"] = round(a["SQ_WAIT_INST_ANY"] / max(a["SQ_WAVE_CYCLES"], 1), 3)
            e["lds_bank_conflict_cycles_per_lds_inst"] = round(a["SQ_LDS_BANK_CONFLICT"] / max(a.get("SQ_INSTS_LDS", 0), 1), 2)
    out[name] = e
out["_note"] = ("separate --pmc passes of `bench.py --steps 1 --warmup 0 --no-cpu --no-extras --size 1024` (2^24 elements); "
                "sq_per_launch: a third pass (8 SQ counters + GRBM_GUI_ACTIVE), valu_wave_insts_per_simd_cycle = SQ_INSTS_VALU / 1024 SIMDs / "
                "(GRBM_GUI_ACTIVE / 8 XCDs); "
                "FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B); msb rounds: "
                "round_pre<first level> reads 32 B of G / P words and writes 64 B of message + 8 B of state + 2 B of backup per element; "
                "round_mid<first level> reads the 64-byte message + 8 B of G1 + 8 B of state and writes the 32-byte message + "
                "4 + 4 B of state, every later level half of the one before")
json.dump(out, open("profiles/r3_pmc_bitwise.json", "w"), indent=1)
for k, v in out.items():
    if k != "_note":
        print(k, {a: (round(b) if isinstance(b, float) else b) for a, b in v.items() if a != "algorithmic"})
