#!/bin/bash
# A/B of kernel builds: benches every _variants/*.so (MWB_LIB override) back to back, twice.  usage: scripts/ab_variants.sh [workload] [extra bench args]
wl=${1:-maze8192}; shift
for rep in 1 2; do
  for so in _variants/*.so; do
    MWB_LIB=$PWD/$so python bench.py --no-cpu-baseline --no-vecenv --workload $wl "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$so', round(d['value']/1e6,3), 'ms/step', round(d['ms_per_step'],4), 'render', round(k['render'],4), 'step', round(k['step'],4), 'prep', round(k['prep'],4), 'reset', round(k['reset'],3))"
  done
done
