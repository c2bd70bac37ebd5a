#!/bin/bash
# rocprofv3 recipe behind profiles/<tag>_<workload>_*: one kernel-trace pass and separate --pmc passes
# (never combined with other trace domains), each with the program directly after `--`.
# usage (on the GPU box, from the repo root): scripts/profile_workload.sh <round_tag> <workload>
set -eo pipefail
TAG=${1:-r01}; WL=${2:-maze8192}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
P=gpurun_out/prof_$WL
rm -rf "$P"
W="--workload $WL --no-cpu-baseline --no-vecenv"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 bench.py $W --steps 100 --warmup 20 > $P.trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -- python3 bench.py $W --steps 20 --warmup 5 > $P.fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -- python3 bench.py $W --steps 20 --warmup 5 > $P.write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $P/pmc_sq -- python3 bench.py $W --steps 20 --warmup 5 > $P.sq.log 2>&1
python3 scripts/summarize_prof.py $P $TAG $WL
mkdir -p gpurun_out/profiles && cp profiles/${TAG}_${WL}_* profiles/${TAG}_pmc_traffic.json gpurun_out/profiles/
