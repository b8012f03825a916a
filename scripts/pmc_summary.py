#!/usr/bin/env python3
"""Summarise the counter_collection CSVs written by scripts/pmc.sh: per counter, the value of
the LAST dispatch of each kernel whose name contains a filter (default: render_kernel<5, false)."""
import csv
import glob
import os
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "render_kernel<5, false"
res = {}
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if flt not in row["Kernel_Name"]:
                continue
            res.setdefault(row["Counter_Name"], {})[int(row["Dispatch_Id"])] = float(row["Counter_Value"])
print(f"# counters of kernels matching {flt!r} in {d}: value of the last dispatch (and number of dispatches seen)")
for k in sorted(res):
    v = res[k]
    last = v[max(v)]
    print(f"{k:28s} {last:22.1f}   ({len(v)} dispatches)")
