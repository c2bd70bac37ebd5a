#!/bin/bash
# round 3, entity tasks in: the whole GPU suite, bench lines of the new workloads + the headline, PMC traffic of pickupobjs8192
set -o pipefail
mkdir -p gpurun_out/r03c
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r03c/gputests.log 2>&1; echo "gpu tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r03c/gputests.log | tail -12
for wl in pickupobjs8192 collecthealth8192 sidewalk8192 maze8192; do
  timeout -k 10 300 python bench.py --workload $wl > gpurun_out/r03c/bench_$wl.json 2> gpurun_out/r03c/bench_$wl.err || echo "bench $wl failed"
  python -c "import json;d=json.load(open('gpurun_out/r03c/bench_$wl.json'));print('$wl',round(d['value']/1e6,3),d['kernel_ms'],d.get('cpu_baseline',{}).get('value'),d.get('vecenv',{}).get('vs_c_abi'))"
done
