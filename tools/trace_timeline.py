#!/usr/bin/env python3
"""One step's kernel timeline from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py: start / end / duration of every
kernel of the LAST complete step (bounded by the stitch kernel that opens each step), per stream, so overlap of the optimizer's side
stream with the backward can be read off.  usage: trace_timeline.py <dir-or-csv> [min_us]"""
import csv
import glob
import os
import sys


def main():
    path = sys.argv[1]
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "stitch" in r["Kernel_Name"]]
    if len(starts) < 3:
        raise SystemExit("fewer than three steps in the trace")
    a, b = starts[-2], starts[-1]
    t0 = int(rows[a]["Start_Timestamp"])
    print(f"# {path}: step of {(int(rows[b]['Start_Timestamp']) - t0) / 1e6:.3f} ms, kernels >= {min_us} us")
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if (e - s) / 1e3 < min_us:
            continue
        name = r["Kernel_Name"]
        name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        print(f"{(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f}  {(e - s) / 1e6:7.3f} ms  q{r.get('Queue_Id', '?')}  vgpr {r.get('VGPR_Count', '?'):>4} acc {r.get('Accum_VGPR_Count', '?'):>4}  {name}")


if __name__ == "__main__":
    main()
