#!/bin/bash
# What bounds reset_kernel when every env times out together (scripts/mass_timeout.py): issue / wait / instruction-cache / HBM counters of its
# largest launches, one --pmc pass per set.  usage: scripts/reset_counters.sh [out-tag] [env_id]
set -eo pipefail
TAG=${1:-r03_reset_pmc}; ENV=${2:-MiniWorld-Maze-v0}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/$TAG
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); P=gpurun_out/$TAG/pass$i
  rm -rf $P
  timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d $P -- python3 scripts/mass_timeout.py $ENV > $P.log 2>&1
  python3 - "$P" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f[0])):
    if "reset_kernel" in r["Kernel_Name"] and "mark" not in r["Kernel_Name"]:
        per[r["Dispatch_Id"]][r["Counter_Name"]] = per[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
# the three mass time-out launches: the largest by the first counter
names = sorted({k for d in per.values() for k in d})
big = sorted(per.values(), key=lambda d: -d.get(names[0], 0))[1:4]   # [0] is the initial reset (4096 blocks)
for n in names: print(n, round(sum(d.get(n, 0) for d in big) / max(1, len(big))))
PY
done
