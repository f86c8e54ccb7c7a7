#!/usr/bin/env python3
"""HBM bytes per launch of any kernels, from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, as the gfx950 guide
prescribes) over the same command.

    python tools/pmc_traffic_any.py FETCH_DIR WRITE_DIR OUT.json 'pattern=algorithmic_bytes[:label]' ...
FETCH_SIZE is in KB and, on gfx950, reports HALF the bytes of a wide coalesced stream (MI355X_MICROARCH.md, HBM): doubled here.
WRITE_SIZE (KB) is exact for 16-byte and dword streaming stores.  Per kernel: the median over launches.
"""
import csv
import glob
import json
import sys


def per_launch(directory, counter, pattern):
    f = sorted(glob.glob(directory + "/**/*counter_collection.csv", recursive=True))[0]
    per, dur = {}, {}
    for r in csv.DictReader(open(f)):
        if pattern in r["Kernel_Name"] and r["Counter_Name"] == counter:
            per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
            dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    vals = sorted(per.values())
    d = sorted(dur.values())
    return (vals[len(vals) // 2] * 1024.0, len(vals), d[len(d) // 2] / 1e6) if vals else (None, 0, None)


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    res = {"correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B); WRITE_SIZE exact",
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)", "kernels": []}
    for spec in sys.argv[4:]:
        pattern, rest = spec.split("=", 1)
        alg, _, label = rest.partition(":")
        fetch, n1, ms1 = per_launch(fetch_dir, "FETCH_SIZE", pattern)
        write, n2, ms2 = per_launch(write_dir, "WRITE_SIZE", pattern)
        if fetch is None or write is None:
            res["kernels"].append({"kernel": pattern, "label": label, "error": "no launches matched"})
            continue
        total = 2.0 * fetch + write
        res["kernels"].append({"kernel": pattern, "label": label, "fetch_bytes": 2.0 * fetch, "write_bytes": write, "hbm_bytes_per_launch": total,
                               "algorithmic_bytes": float(alg), "traffic_over_algorithmic": round(total / float(alg), 3),
                               "launches": [n1, n2], "duration_ms_under_profiler": [round(ms1, 3), round(ms2, 3)],
                               "hbm_GBs": round(total / (ms1 * 1e-3) / 1e9, 1)})
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
