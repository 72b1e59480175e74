#!/usr/bin/env python
"""The last launches of k_scan_step in a rocprofv3 --kernel-trace CSV as a timeline (start, end, workgroups, queue), relative to the
first of them: what the fill and the drain of a short timed run look like. Usage: python scripts/timeline_tail.py <trace.csv> [n [all]]"""
import csv
import sys

every = len(sys.argv) > 3 and sys.argv[3] == "all"  # all kernels of the library, not k_scan_step alone
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_scan_step" in r["Kernel_Name"] or (every and "ss::" in r["Kernel_Name"])]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s / 1e3:9.1f} {e / 1e3:9.1f}  dur {(e - s) / 1e3:6.1f}  wgs {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):5d}  queue {r.get('Queue_Id', '?')}" + ("" if "k_scan_step" in r["Kernel_Name"] else "  " + r["Kernel_Name"].split("(")[0]))
