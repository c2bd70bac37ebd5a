set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT[A-Z_]*\|SQ_INSTS_BRANCH\|SQ_INSTS_SMEM\|SQ_ACTIVE_INST[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*" | sort -u > gpurun_out/avail_counters.txt || true
P=gpurun_out/icache
rm -rf $P
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $P -- python3 bench.py --workload maze8192 --no-cpu-baseline --no-vecenv --steps 20 --warmup 5 > $P.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/icache/*/*_counter_collection.csv")
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "render_kernel<256, 2" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()): print(k, sum(v)/len(v))
PY
