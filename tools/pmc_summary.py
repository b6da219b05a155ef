#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, sum of each counter."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in rows:
            k = r["Kernel_Name"].split("(")[0][:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
        for k, v in agg.items():
            if "demod" not in k:
                continue
            print(f"{d}: {k}  dispatches={len(disp[k])}")
            for c, val in sorted(v.items()):
                print(f"    {c:28s} {val:16.0f}")
