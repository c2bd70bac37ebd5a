#!/bin/bash
# SQ issue-side counters of the bulk render kernel (who occupies the issue ports): one --pmc pass.  usage: scripts/issue_counters.sh [workload]
set -eo pipefail
WL=${1:-maze8192}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for SET in "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  P=gpurun_out/issue_$(echo $SET | cut -c1-20 | tr ' ' '_')
  rm -rf $P
  timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $P -- python3 bench.py --workload $WL --no-cpu-baseline --no-vecenv --steps 20 --warmup 5 > $P.log 2>&1
  python3 - "$P" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "render_kernel<256, 2" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()): print(k, round(sum(v)/len(v)))
PY
done
