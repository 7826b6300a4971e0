#!/usr/bin/env python3
"""Print the kernel sequence of the LAST frame in a rocprofv3 kernel_trace.csv (name, duration us, gap to previous us)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# frames start with k_primary
starts = [i for i, r in enumerate(rows) if "k_primary<false>" in r["Kernel_Name"]]
ends = [i for i, r in enumerate(rows) if "k_primary" in r["Kernel_Name"]]
a = starts[-1]
prev_end = None
t0 = int(rows[a]["Start_Timestamp"])
stop = min([i for i in ends if i > a] + [len(rows)])
for r in rows[a:stop]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void flx::", "")
    print("%-34s start %8.1f us  dur %8.1f us  gap %6.1f us" % (name[:34], (s - t0) / 1e3, (e - s) / 1e3, 0 if prev_end is None else (s - prev_end) / 1e3))
    prev_end = e
