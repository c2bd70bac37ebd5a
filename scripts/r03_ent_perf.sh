#!/bin/bash
# entity render kernel experiments: parity of the entity tasks, then the three entity workloads (short runs, no CPU legs)
out=gpurun_out/${1:-r03_entperf}; mkdir -p $out
if [ -z "$SKIP_TESTS" ]; then timeout -k 10 600 python -m pytest tests/test_gpu_ents.py tests/test_gpu_view.py -q -x > $out/tests.log 2>&1; echo "tests rc=$?"; tail -2 $out/tests.log; fi
for wl in pickupobjs8192 collecthealth8192 sidewalk8192; do
  timeout -k 10 200 python bench.py --workload $wl --steps 60 --warmup 20 --no-vecenv --no-cpu-baseline > $out/bench_$wl.json 2>/dev/null
  python -c "import json;d=json.load(open('$out/bench_$wl.json'));print('$wl',round(d['value']/1e6,3),round(d['kernel_ms']['render'],3))"
done
shift
for alt in "$@"; do   # further arguments: environment settings to time the same three workloads under
  for wl in pickupobjs8192 collecthealth8192 sidewalk8192; do
    env $alt timeout -k 10 200 python bench.py --workload $wl --steps 60 --warmup 20 --no-vecenv --no-cpu-baseline > $out/bench_${wl}_$alt.json 2>/dev/null
    python -c "import json;d=json.load(open('$out/bench_${wl}_$alt.json'));print('$alt $wl',round(d['value']/1e6,3),round(d['kernel_ms']['render'],3))"
  done
done
