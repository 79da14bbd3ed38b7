#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: mean duration per (kernel, grid size)."""
import csv
import glob
import sys
from collections import defaultdict

path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
acc = defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "sr_" not in name:
            continue
        short = name.split("sr_")[-1 if "sr_conv_impl" in name else 1][:44]  # sr_conv_impl::sr_conv3x3_kernel<...> -> conv3x3_kernel<...>
        acc[(short, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(acc):
    v = sorted(acc[k])
    print(f"{k[0]:46s} grid=({k[1]:5d},{k[2]:2d}) n={len(v):4d} med={v[len(v)//2]/1e3:9.1f}us min={v[0]/1e3:9.1f}us")
