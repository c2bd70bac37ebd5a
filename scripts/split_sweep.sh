#!/bin/bash
# MWB_SPLIT sweep: how many of the cheapest envs a bulk render launch draws as two half-frame workgroups.  usage: scripts/split_sweep.sh [workload]
wl=${1:-maze8192}
for sp in 0 384 768 1152 1536 2048; do
  MWB_SPLIT=$sp python bench.py --no-cpu-baseline --no-vecenv --workload $wl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl split', $sp, round(d['value']/1e6,3), round(d['kernel_ms']['render'],4))"
done
