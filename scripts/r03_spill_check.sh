#!/bin/bash
# round 3, after the spill removal: GPU tests, bench lines of the workloads whose kernels changed, PMC traffic of YMaze / PutNext
set -o pipefail
mkdir -p gpurun_out/r03a
python -m pytest tests -m gpu -x -q > gpurun_out/r03a/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r03a/gputests.log
for wl in ymaze8192 putnext8192 tmaze_features8192 maze8192; do
  python bench.py --workload $wl --no-cpu-baseline > gpurun_out/r03a/bench_$wl.json 2> gpurun_out/r03a/bench_$wl.err || echo "bench $wl failed"
  python -c "import json;d=json.load(open('gpurun_out/r03a/bench_$wl.json'));print('$wl',round(d['value']/1e6,3),d['kernel_ms'])"
done
for wl in ymaze8192 putnext8192; do
  bash scripts/profile_workload.sh r03 $wl > gpurun_out/r03a/prof_$wl.log 2>&1 || echo "profile $wl failed"
  tail -1 gpurun_out/r03a/prof_$wl.log | cut -c1-300
done
