#!/usr/bin/env python3
"""Prints three steady-state steps of a rocprofv3 --kernel-trace run as a timeline (start us, duration us, queue, kernel).
usage: timeline_print.py <dir with *_kernel_trace.csv> [first step index]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Queue_Id"])))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("step_kernel")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) * 2 // 3
i0, i1 = starts[k], starts[k + 3]
t0 = rows[i0][0]
for r in rows[i0:i1]:
    print("%8.1f %8.1f  q%d  %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[2][:44]))
