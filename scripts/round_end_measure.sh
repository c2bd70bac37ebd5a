#!/bin/bash
# everything profiles/<tag>_* is made of, in one GPU call: bench lines of all workloads, the driver's protocol,
# rocprofv3 summaries of three workloads, the render parity soak.  usage (GPU box, repo root): scripts/round_end_measure.sh <tag>
set -o pipefail
TAG=${1:-r02}
mkdir -p gpurun_out/final
for wl in maze8192 oneroom4096 maze8192_depth fourrooms16384_dr tmaze_features8192 sim2real_push8192 putnext8192 ymaze8192; do
  python bench.py --workload $wl > gpurun_out/final/${TAG}_bench_$wl.json 2> gpurun_out/final/bench_$wl.err || echo "bench $wl failed"
  echo "bench $wl done"
done
bash scripts/driver_protocol.sh > gpurun_out/final/${TAG}_driver_protocol.txt 2>&1
echo "driver protocol done"
for wl in maze8192 oneroom4096 ymaze8192; do
  bash scripts/profile_workload.sh $TAG $wl > gpurun_out/final/prof_$wl.log 2>&1 || echo "profile $wl failed"
  echo "profile $wl done"
done
python scripts/parity_soak.py 1536 > gpurun_out/final/${TAG}_parity_soak.txt 2>&1
echo "soak done"
