#!/bin/bash
# SQ instruction counters of the render kernel per phase: MWB_DEBUG 0 = all, 2 = no 8-sample path,
# 4 = no interior shading, 6 = corner classification only, 70 = 6 without the per-pixel corner passes, 134 = prologue + copy-out.  usage: scripts/phase_counters.sh [workload]
set -eo pipefail
WL=${1:-maze8192}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/phase
for D in 0 2 4 6 70 134; do
  export MWB_DEBUG=$D
  P=gpurun_out/phase/d$D
  rm -rf $P
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $P -- python3 bench.py --workload $WL --no-cpu-baseline --no-vecenv --steps 20 --warmup 5 > $P.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for D in (0, 2, 4, 6, 70, 134):
    f = glob.glob("gpurun_out/phase/d%d/*/*_counter_collection.csv" % D)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "render_kernel<256, 2" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[D] = {k: sum(v) / len(v) for k, v in agg.items()}
    b = [l for l in open("gpurun_out/phase/d%d.log" % D) if l.startswith("{")]
    if b:
        out[D]["render_ms"] = json.loads(b[-1])["kernel_ms"]["render"]
json.dump(out, open("gpurun_out/phase/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
