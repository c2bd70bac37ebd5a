#!/bin/bash
# A/B of an environment switch read at mwb_create.  usage: scripts/ab_env.sh VAR=VALUE [workloads...]
var=$1; shift
for wl in ${@:-maze8192}; do for rep in 1 2; do for v in "MWB_AB_DUMMY=1" "$var"; do
  env $v python bench.py --no-cpu-baseline --no-vecenv --workload $wl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$wl', '$v', round(d['value']/1e6,3), 'ms/step', round(d['ms_per_step'],4), 'render', round(k['render'],4), 'step', round(k['step'],4), 'prep', round(k['prep'],4), 'reset', round(k['reset'],3))"
done; done; done
