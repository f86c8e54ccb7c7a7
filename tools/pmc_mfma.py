#!/usr/bin/env python3
"""MFMA utilisation and effective clock per kernel from ONE rocprofv3 PMC pass of bench.py
(--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE, kernel-trace on for the durations).

    python tools/pmc_mfma.py gpurun_out/pmc_mfma profiles/r01_mfma_util.json
SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles in which a SIMD's matrix pipe is busy, summed over all SIMDs
(v_mfma_f32_32x32x2_f32 = 64 per instruction); GRBM_GUI_ACTIVE is reported summed over the 8 XCDs.
util = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs); clock = GUI_ACTIVE / 8 / duration (MI355X_MICROARCH.md, DVFS)."""
import collections
import csv
import glob
import json
import sys


def main():
    d, out = sys.argv[1:3]
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    meta = {}
    for r in csv.DictReader(open(f)):
        key = r["Dispatch_Id"]
        per[key][r["Counter_Name"]] += float(r["Counter_Value"])
        meta[key] = (r["Kernel_Name"], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    agg = collections.defaultdict(list)
    for key, c in per.items():
        name, ns = meta[key]
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if gui <= 0:
            continue
        agg[name].append((c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024.0), gui / ns if ns > 0 else 0.0, ns))
    rows = []
    for name, v in agg.items():
        v.sort()
        util, ghz, ns = v[len(v) // 2]
        if util > 0.01:
            rows.append({"kernel": name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80], "launches": len(v), "mfma_util": round(util, 4),
                         "clock_ghz": round(ghz, 3), "duration_us": round(ns / 1e3, 1)})
    rows.sort(key=lambda r: -r["duration_us"])
    res = {"source": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python bench.py",
           "definition": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); medians over launches; "
                         "profiled passes run at a lower clock than un-profiled ones",
           "kernels": rows}
    json.dump(res, open(out, "w"), indent=1)
    for r in rows:
        print(r)


if __name__ == "__main__":
    main()
