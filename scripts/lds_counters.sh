cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_LDS[A-Z_]*\|SQ_ACTIVE_INST_LDS\|SQ_INSTS_LDS" | sort -u | tr '\n' ' '; echo
P=gpurun_out/ldsc; rm -rf $P
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $P -- python3 bench.py --workload maze8192 --no-cpu-baseline --no-vecenv --steps 20 --warmup 5 > $P.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/ldsc/*/*_counter_collection.csv")
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "render_kernel<256, 2" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()): print(k, round(sum(v)/len(v)))
PY
tail -3 $P.log
