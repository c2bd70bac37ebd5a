#!/bin/bash
# the driver's round-end command, three times (run-to-run spread), then the default (300 / 50) run.  usage: scripts/driver_protocol.sh
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('20/5', round(d['value']/1e6,3), round(d['ms_per_step'],4), round(d['kernel_ms']['render'],4), d['vecenv']['vs_c_abi'] if 'vecenv' in d else None)"
done
python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('300/50', round(d['value']/1e6,3), round(d['ms_per_step'],4), round(d['kernel_ms']['render'],4), d['vecenv']['vs_c_abi'] if 'vecenv' in d else None)"
