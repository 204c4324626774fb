#!/usr/bin/env python3
"""Kernel timeline of the LAST MSM of a rocprofv3 --kernel-trace run of tools/prof_msm.py: per launch name, start offset, duration and the gap to
the previous kernel's end (us).   timeline_msm.py <rocprofv3 output dir>      (reads the rocpd database or the csv kernel trace)"""
import csv, glob, os, sqlite3, sys
rows = []
dbs = glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)
if dbs:
    rows = [(int(s), int(e), nm, gx, wx) for s, e, nm, gx, wx in sqlite3.connect(dbs[0]).execute("select start, end, name, grid_x, workgroup_x from kernels")]
else:
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", 0) or 0), int(r.get("Workgroup_Size_X", 0) or 0)))
rows.sort()
last = max(i for i, r in enumerate(rows) if "k_digits" in r[2])
t0, prev = rows[last][0], rows[last][0]
for s, e, nm, gx, wx in rows[last:]:
    short = nm.split("(")[0].replace("void ", "").replace("zkhip::", "")
    print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:8.2f}  gap {(s - prev) / 1e3:6.2f}  {short}  grid {gx} x {wx}")
    prev = e
print(f"total {(prev - t0) / 1e3:.2f} us")
