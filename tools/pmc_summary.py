#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv: mean per (kernel, grid) of every counter."""
import csv
import glob
import sys
from collections import defaultdict

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "sr_" not in name:
            continue
        key = (name.split("sr_")[-1 if "sr_conv_impl" in name else 1][:40], int(r["Grid_Size"]) // int(r["Workgroup_Size"]) if "Grid_Size" in r else 0)
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"    {c:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
