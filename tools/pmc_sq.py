#!/usr/bin/env python3
"""Per-kernel medians of the SQ counters of one or more rocprofv3 PMC passes (kernel-trace on for the durations).

    python tools/pmc_sq.py OUT.json DIR [DIR ...]
Every DIR holds the *_counter_collection.csv of one `rocprofv3 --kernel-trace --pmc ... --output-format csv` pass over the
same command (separate passes for counter groups that do not fit the 8 SQ slots).  Per kernel and counter: the median
over launches of the counter summed over its dimensions.  Derived, where the inputs exist:
  mfma_busy       = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)        share of all SIMD cycles with the matrix pipe busy
  coexec_of_mfma  = SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES              share of matrix-busy cycles with a VALU op executing beside it
  valu_per_mfma_cycle64 = SQ_INSTS_VALU / (SQ_VALU_MFMA_BUSY_CYCLES / 64)              vector instructions per 64-cycle fp32 MFMA
  clock_ghz       = GRBM_GUI_ACTIVE / 8 / duration
(SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles: MI355X_MICROARCH.md.)"""
import collections
import csv
import glob
import json
import sys


def load(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        raise SystemExit(f"{d}: no counter_collection.csv")
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    meta = {}
    for r in csv.DictReader(open(f[0])):
        key = r["Dispatch_Id"]
        per[key][r["Counter_Name"]] += float(r["Counter_Value"])
        meta[key] = (r["Kernel_Name"], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for key, c in per.items():
        name, ns = meta[key]
        name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
        for k, v in c.items():
            agg[name][k].append(v)
        agg[name]["duration_ns"].append(ns)
    return agg


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    merged = collections.defaultdict(dict)
    for d in dirs:
        for name, c in load(d).items():
            for k, v in c.items():
                if k == "duration_ns" and k in merged[name]:
                    continue
                merged[name][k] = med(v)
            merged[name]["launches"] = max(merged[name].get("launches", 0), len(c["duration_ns"]))
    rows = []
    for name, c in merged.items():
        if c.get("duration_ns", 0) < 2e5:
            continue
        r = {"kernel": name, "launches": c["launches"], "duration_us": round(c["duration_ns"] / 1e3, 1)}
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if gui > 0:
            r["clock_ghz"] = round(gui / c["duration_ns"], 3)
            if busy:
                r["mfma_busy"] = round(busy / (gui * 1024.0), 4)
        if busy and "SQ_VALU_MFMA_COEXEC_CYCLES" in c:
            r["coexec_of_mfma"] = round(c["SQ_VALU_MFMA_COEXEC_CYCLES"] / busy, 4)
        if busy and "SQ_INSTS_VALU" in c:
            r["valu_per_mfma_cycle64"] = round(c["SQ_INSTS_VALU"] / (busy / 64.0), 3)
        for k in sorted(c):
            if k.startswith("SQ_") or k.startswith("GRBM_"):
                r[k] = c[k]
        rows.append(r)
    rows.sort(key=lambda r: -r["duration_us"])
    json.dump({"source": "rocprofv3 --kernel-trace --pmc <counters> --output-format csv; medians over launches; profiled passes run at a "
                         "lower clock than un-profiled ones", "kernels": rows}, open(out, "w"), indent=1)
    for r in rows:
        print({k: v for k, v in r.items() if not (k.startswith("SQ_") or k.startswith("GRBM_"))})


if __name__ == "__main__":
    main()
