#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per kernel per counter."""
import csv, glob, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else "a"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/p_counter_collection.csv")):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0].replace("void edt_amd::", "")[:64]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, ctrs in acc.items():
    print(k)
    for c, v in sorted(ctrs.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
