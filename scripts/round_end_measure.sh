#!/bin/bash
# everything profiles/<tag>_* is made of, in two GPU calls (a call is limited to 20 minutes):
#   part 1: bench lines of all workloads, the driver's protocol, the mass time-out step
#   part 2: rocprofv3 summaries of four workloads, the render parity soak
# usage (GPU box, repo root): scripts/round_end_measure.sh <tag> <1|2>
set -o pipefail
TAG=${1:-r03}; PART=${2:-1}
mkdir -p gpurun_out/final
if [ "$PART" = "1" ]; then
  for wl in maze8192 oneroom4096 maze8192_depth fourrooms16384_dr tmaze_features8192 sim2real_push8192 putnext8192 ymaze8192 pickupobjs8192 collecthealth8192 sidewalk8192; do
    timeout -k 10 240 python bench.py --workload $wl > gpurun_out/final/${TAG}_bench_$wl.json 2> gpurun_out/final/bench_$wl.err || echo "bench $wl failed"
    echo "bench $wl done"
  done
  bash scripts/driver_protocol.sh > gpurun_out/final/${TAG}_driver_protocol.txt 2>&1
  echo "driver protocol done"
  timeout -k 10 120 python scripts/mass_timeout.py > gpurun_out/final/${TAG}_mass_timeout.txt 2>&1
  timeout -k 10 120 python scripts/mass_timeout.py MiniWorld-FourRooms-v0 >> gpurun_out/final/${TAG}_mass_timeout.txt 2>&1
  timeout -k 10 200 python bench.py --force-gather --steps 100 --warmup 30 --no-vecenv --no-cpu-baseline > gpurun_out/final/${TAG}_bench_forcegather.json 2>/dev/null
  echo "part 1 done"
else
  for wl in maze8192 oneroom4096 ymaze8192 pickupobjs8192; do
    bash scripts/profile_workload.sh $TAG $wl > gpurun_out/final/prof_$wl.log 2>&1 || echo "profile $wl failed"
    echo "profile $wl done"
  done
  timeout -k 10 400 python scripts/parity_soak.py 1536 > gpurun_out/final/${TAG}_parity_soak.txt 2>&1
  echo "soak done"
fi
