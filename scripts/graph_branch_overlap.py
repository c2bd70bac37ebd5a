#!/usr/bin/env python3
"""Where does a replayed hipGraph of mwb_step put the side chain (reset_kernel -> the regenerated envs' render) relative to the bulk render?
Reads a rocprofv3 kernel trace of scripts/ab_graph_replay.py (eager, graph, eager, graph in one process) and prints, per phase, how
the side chain's kernels lie against the bulk render of the same step.  usage: graph_branch_overlap.py <kernel_trace.csv>"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    kind = "bulk" if "render_kernel<256, 2" in n else "reset" if n.startswith("void reset_kernel") else "side" if "render_kernel<256, 1" in n else "step" if "step_kernel" in n else None
    if kind:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
rows.sort()
# steps = from one step_kernel to the next
steps, cur = [], None
for s, e, k in rows:
    if k == "step":
        if cur:
            steps.append(cur)
        cur = {"t0": s}
    elif cur is not None:
        cur.setdefault(k, (s, e))
if cur:
    steps.append(cur)
steps = [st for st in steps if all(k in st for k in ("bulk", "reset", "side"))]
# phases of 350 steps each (50 warm-up + 300 timed), as ab_graph_replay.py runs them
per = len(steps) // 4
for ph, name in enumerate(("eager 1", "graph 1", "eager 2", "graph 2")):
    seg = steps[ph * per + 60:(ph + 1) * per - 5]
    if not seg:
        continue
    lead = [(st["bulk"][0] - st["reset"][0]) / 1e3 for st in seg]          # > 0: the reset started before the bulk render
    tail = [(st["side"][1] - st["bulk"][1]) / 1e3 for st in seg]           # > 0: the side chain ended after the bulk render
    dur = [(max(st["side"][1], st["bulk"][1]) - st["t0"]) / 1e3 for st in seg]
    bulk = [(st["bulk"][1] - st["bulk"][0]) / 1e3 for st in seg]
    m = lambda v: sum(v) / len(v)   # noqa: E731
    print("%-8s steps %3d  reset starts %+7.1f us before the bulk render   side chain ends %+7.1f us after it   bulk render %6.1f us   step %6.1f us"
          % (name, len(seg), m(lead), m(tail), m(bulk), m(dur)))
