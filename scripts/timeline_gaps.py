#!/usr/bin/env python3
"""Where does a step's wall time go besides the bulk render?  Reads a rocprofv3 --kernel-trace CSV of a bench run and
prints, for the steady-state steps, the mean duration of every kernel and the idle gaps on the critical path
(end of one main-stream kernel -> start of the next).  usage: timeline_gaps.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Queue_Id"])))
rows.sort()
# steps = from one step_kernel start to the next
starts = [i for i, r in enumerate(rows) if r[2].startswith("step_kernel")]
steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
steps = steps[len(steps) // 3:]   # steady state
acc, gaps, total = {}, {}, 0.0
for st in steps:
    total += (st[-1][1] if False else 0)
    main = [r for r in st if r[2].startswith(("step_kernel", "prep_kernel", "render_kernel<256, 2")) or "render_kernelILi256ELi2" in r[2]]
    for r in st:
        acc.setdefault(r[2][:60], []).append((r[1] - r[0]) / 1e3)
    for a, b in zip(main[:-1], main[1:]):
        gaps.setdefault(a[2][:28] + " -> " + b[2][:28], []).append((b[0] - a[1]) / 1e3)
span = [(b[0][0] - a[0][0]) / 1e3 for a, b in zip(steps[:-1], steps[1:])]
print("steps analysed %d, mean step period %.1f us" % (len(steps), sum(span) / max(1, len(span))))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("  %-60s n/step %.2f  mean %.1f us" % (k, len(v) / len(steps), sum(v) / len(v)))
for k, v in gaps.items():
    print("  gap %-60s mean %.1f us" % (k, sum(v) / len(v)))
# gap between the last main kernel of a step and the next step_kernel
tail = [(b[0][0] - max(r[1] for r in a if not r[2].startswith(("reset", "order", "clear")) or True)) / 1e3 for a, b in zip(steps[:-1], steps[1:])]
print("  gap end-of-step (latest kernel end on any stream) -> next step_kernel: mean %.1f us" % (sum(tail) / max(1, len(tail))))
