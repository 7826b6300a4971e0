#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, mean of each counter over dispatches."""
import csv
import collections
import sys

for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("#", path)
    for name, ctrs in acc.items():
        if "rocclr" in name or "at::" in name:
            continue
        print(name, "dispatches", max(len(v) for v in ctrs.values()))
        for c, v in sorted(ctrs.items()):
            big = [x for x in v if x >= 0.5 * max(v)]       # bench.py launches batches of frames and single frames: the batches on their own
            print("   %-28s %.4g   (the %d largest launches: %.4g)" % (c, sum(v) / len(v), len(big), sum(big) / len(big)))
