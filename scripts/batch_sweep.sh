#!/bin/bash
# env-steps/s and bulk render time against the batch size (envs per GPU).  usage: scripts/batch_sweep.sh [workload]
wl=${1:-maze8192}
for n in 1024 2048 4096 8192 12288 16384 32768; do
  python bench.py --no-cpu-baseline --no-vecenv --workload $wl --envs-per-gpu $n --steps 200 --warmup 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl envs', $n, round(d['value']/1e6,3), 'M env-steps/s  ms/step', round(d['ms_per_step'],4), 'render', round(d['kernel_ms']['render'],4))"
done
