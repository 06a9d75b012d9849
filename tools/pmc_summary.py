#!/usr/bin/env python3
"""Mean counter values per kernel from rocprofv3 counter_collection.csv files: pmc_summary.py <kernel-substring> <dir>..."""
import csv, glob, sys, collections
key = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
