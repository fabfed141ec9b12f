#!/usr/bin/env python3
"""Print the counters of every dispatch of kernels matching a substring from a rocprofv3 --pmc CSV directory.

    python tools/pmc_kernel.py gpurun_out/pmc_dir attn_kernel
"""
import collections
import csv
import glob
import os
import sys

src, pat = sys.argv[1], sys.argv[2]
for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if pat in r["Kernel_Name"]:
            d = rows.setdefault(r["Dispatch_Id"], {"dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for k, d in list(rows.items())[-2:]:
        print(f"dispatch {k}: " + "  ".join(f"{n}={v:.4g}" for n, v in d.items()))
