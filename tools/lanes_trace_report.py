#!/usr/bin/env python3
"""kernel_trace.csv of tools/lanes_trace.py -> per kernel start / end (ms since the first) of the last frames, and how much of the k_wf_frame
launches overlap each other."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
tail = rows[-40:]
for r in tail:
    print("%-28s queue %-3s %10.3f .. %10.3f ms" % (r["Kernel_Name"].split("(")[0].replace("void ", "").replace("flx::", "")[:28], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6))
fr = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_wf_frame" in r["Kernel_Name"]][-16:]
ov = sum(max(0, min(a[1], b[1]) - max(a[0], b[0])) for a, b in zip(fr, fr[1:]))
print("k_wf_frame: %d launches, mean %.3f ms, overlap between consecutive launches %.3f ms on average" % (len(fr), sum(e - s for s, e in fr) / len(fr) / 1e6, ov / max(1, len(fr) - 1) / 1e6))
