#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof/{trace,pmc_fetch,pmc_write,pmc_sq}) into the small,
tracked summaries under profiles/.  usage: summarize_prof.py <prof_dir> <round_tag> <workload>

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are
collected in separate --pmc passes, are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes
of wide (16 B/lane) coalesced reads, so the read side is doubled (an upper bound for the scattered
4-byte texel fetches, which the guide calls uncalibrated)."""
import collections
import csv
import glob
import json
import os
import sys

prof, tag, workload = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(root, "profiles")
os.makedirs(out_dir, exist_ok=True)

stats = list(csv.DictReader(open(max(glob.glob(os.path.join(prof, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime))))
ours = [r for r in stats if any(k in r["Name"] for k in ("render_kernel", "reset_kernel", "step_kernel", "prep_kernel", "env_kernel"))]
with open(os.path.join(out_dir, "%s_%s_kernel_stats.csv" % (tag, workload)), "w") as fh:
    w = csv.DictWriter(fh, fieldnames=list(stats[0].keys()))
    w.writeheader()
    for r in ours:
        w.writerow(r)

def counters(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    files = glob.glob(os.path.join(prof, sub, "*", "*_counter_collection.csv"))
    if not files:
        return {}
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    # drop each kernel's first dispatch (the all-env reset / cold caches) when there are several
    return {k: {c: (sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0]) for c, v in cs.items()} for k, cs in agg.items()
            if "kernel" in k and "at::" not in k}

summary = {"workload": workload, "fetch": counters("pmc_fetch"), "write": counters("pmc_write"), "sq": counters("pmc_sq")}
def pick_render(keys):
    import re
    ks = [k for k in keys if "render_kernel" in k]
    for want in ("2", "0"):   # the bulk launch (MODE 2) if the run overlapped reset, else MODE 0
        for k in ks:   # render_kernel<THREADS, MODE[, NBOX]>
            m = re.search(r"render_kernel<\s*\d+\s*,\s*(\d+)", k)
            if m and m.group(1) == want:
                return k
    return ks[0] if ks else None
rk = pick_render(summary["fetch"])
traffic = None
if rk:
    f_kib = summary["fetch"][rk]["FETCH_SIZE"]
    w_kib = summary["write"][rk]["WRITE_SIZE"]
    traffic = int(2 * f_kib * 1024 + w_kib * 1024)
    summary["render_hbm_bytes_per_launch"] = {"FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib, "fetch_x2_plus_write_bytes": traffic}
with open(os.path.join(out_dir, "%s_%s_pmc_summary.json" % (tag, workload)), "w") as fh:
    json.dump(summary, fh, indent=1, sort_keys=True)
tpath = os.path.join(out_dir, "%s_pmc_traffic.json" % tag)
t = json.load(open(tpath)) if os.path.exists(tpath) else {}
t[workload] = {"hbm_bytes_per_render_launch": traffic, "source": "%s_%s_pmc_summary.json" % (tag, workload)}
json.dump(t, open(tpath, "w"), indent=1, sort_keys=True)
print(json.dumps(summary.get("render_hbm_bytes_per_launch")), [ (r["Name"][:30], r["AverageNs"]) for r in ours])
