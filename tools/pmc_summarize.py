"""Summarise tools/pmc_round5.sh: per entry, the counters of the LAST n dispatches of the named kernel (= tools/one_launch.py's replays)."""
import csv
import glob
import json
import os
import sys

out, n = sys.argv[1], int(sys.argv[2])
entries = [l.strip().split("|") for l in open(os.path.join(out, "entries.txt")) if l.strip()]
res = {}
for tag, kern in entries:
    c = {}
    try:
        c["launch"] = open(os.path.join(out, tag + ".info")).read().strip().replace("ONE_LAUNCH ", "")
    except OSError:
        c["launch"] = "?"
    for p in "abcd":
        for f in glob.glob(os.path.join(out, f"{tag}_{p}", "*", "*counter_collection.csv")):
            rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
            if not rows:
                continue
            ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-n:]
            keep = set(ids)
            agg = {}
            for r in rows:
                if int(r["Dispatch_Id"]) in keep:
                    a = agg.setdefault(r["Counter_Name"], [0, 0.0])
                    a[0] += 1
                    a[1] += float(r["Counter_Value"])
                    c["kernel"] = r["Kernel_Name"][:70]
            for k, (cnt, v) in agg.items():
                c[k] = v / cnt
    m = c.get("SQ_INSTS_MFMA")
    if m:
        for k in ("VALU", "SALU", "LDS", "VMEM"):
            if "SQ_INSTS_" + k in c:
                c[k.lower() + "_per_mfma"] = round((c["SQ_INSTS_" + k] - (m if k == "VALU" else 0)) / m, 3)      # (SQ_INSTS_VALU counts the MFMAs too)
    if "GRBM_GUI_ACTIVE" in c:
        c["kernel_cycles"] = c["GRBM_GUI_ACTIVE"] / 8            # (summed over the 8 XCDs)
        if m:
            c["mfma_pipe_busy_frac"] = round(m * 64 / (c["kernel_cycles"] * 1024), 4)      # 1024 SIMDs; an fp32 32x32x2 MFMA holds its SIMD's pipe 64 cycles
    if "SQ_WAVE_CYCLES" in c:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT"):
            if k in c:
                c[k.lower() + "_share"] = round(c[k] / c["SQ_WAVE_CYCLES"], 4)
    if "FETCH_SIZE" in c:
        c["hbm_read_MB_x2"] = round(2 * c["FETCH_SIZE"] / 1024, 2)
    if "WRITE_SIZE" in c:
        c["hbm_write_MB"] = round(c["WRITE_SIZE"] / 1024, 2)
    if "TCC_HIT_sum" in c:
        c["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    res[tag] = c
json.dump(res, open(os.path.join(out, "counters.json"), "w"), indent=1, sort_keys=True)
for tag, c in res.items():
    print(tag)
    print("   ", c.get("launch"))
    print("   ", c.get("kernel"))
    print("   ", {k: c[k] for k in ("kernel_cycles", "mfma_pipe_busy_frac", "valu_per_mfma", "salu_per_mfma", "lds_per_mfma", "vmem_per_mfma", "sq_wait_any_share",
                                    "sq_lds_bank_conflict_share", "hbm_read_MB_x2", "hbm_write_MB", "l2_hit_rate") if k in c})
