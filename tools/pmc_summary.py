#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch of a kernel."""
import csv, glob, sys, collections
kern = sys.argv[1]
for path in sys.argv[2:]:
    for f in (glob.glob(path + "/*/*counter_collection.csv") + glob.glob(path + "/*counter_collection.csv")):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if kern in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
