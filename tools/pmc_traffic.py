#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as the gfx950 guide prescribes) of
tools/bench_kernels.py into per-launch HBM bytes of the dominant kernel and store them under profiles/.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_c2_fwd_traffic.json
FETCH_SIZE is in KB and, on gfx950, reports HALF the bytes of a wide coalesced stream (MI355X_MICROARCH.md, HBM):
it is doubled here.  WRITE_SIZE (KB) is exact for 16-byte and dword streaming stores.
"""
import csv
import glob
import json
import sys

KERNEL = "conv_wino2r_fwd<5, 4"               # c2 forward (Winograd F(2x2,3x3), bias+ReLU+sign bits, 4 waves)


def per_launch(directory, counter, kernel=None):
    kernel = kernel or KERNEL
    f = sorted(glob.glob(directory + "/**/p_counter_collection.csv", recursive=True) or glob.glob(directory + "/**/*counter_collection.csv", recursive=True))[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    vals = sorted(per.values())
    return vals[len(vals) // 2] * 1024.0, len(vals)


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    px = 32 * 256 * 1836
    if len(sys.argv) > 4 and sys.argv[4] == "dgrad_w1":
        # c2 data gradient + c1 weight gradient in one kernel: reads g2 once, one sign word and one NHWC4 image pixel per pixel;
        # writes only the per-wave 32 x 32 partials of dW1
        kernel, algorithmic = "conv_wino2r_fwd<9, 4", px * (32 * 4 + 4 + 16) + 1024 * 1024 * 4
    else:
        kernel, algorithmic = KERNEL, px * (32 * 4 * 2 + 4)    # read a1 once + write a2 once + one sign word per pixel, bs = 32
    fetch, n1 = per_launch(fetch_dir, "FETCH_SIZE", kernel)
    write, n2 = per_launch(write_dir, "WRITE_SIZE", kernel)
    res = {"kernel": kernel, "batch": 32, "fetch_bytes": 2.0 * fetch, "write_bytes": write,
           "hbm_bytes_per_launch": 2.0 * fetch + write, "algorithmic_bytes": algorithmic,
           "launches": [n1, n2], "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B); WRITE_SIZE exact",
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/bench_one.py wino2_fwd,wino2_dgrad_w1"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
