#!/bin/bash
# kernel trace of the mass time-out step (scripts/mass_timeout.py): how long reset_kernel and the side chain's render take there
out=gpurun_out/${1:-r03_mass}; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o mass -- python3 scripts/mass_timeout.py ${2:-MiniWorld-Maze-v0} > $out/prof.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/prof/**/*kernel_trace.csv", recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)): d[r["Kernel_Name"][:44]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items(): print("%-46s n=%4d mean %9.1f us  top %s"%(k, len(v), sum(v)/len(v), [round(x) for x in sorted(v)[-4:]]))
PY
