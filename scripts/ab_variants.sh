#!/bin/bash
# A/B of kernel builds: benches every _variants/*.so (MWB_LIB override) back to back, twice.  usage: scripts/ab_variants.sh [workload]
wl=${1:-maze8192}
for rep in 1 2; do
  for so in _variants/*.so; do
    MWB_LIB=$PWD/$so python bench.py --no-cpu-baseline --no-vecenv --workload $wl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$so', round(d['value']/1e6,3), round(d['kernel_ms']['render'],4))"
  done
done
