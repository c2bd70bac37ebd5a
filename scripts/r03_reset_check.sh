#!/bin/bash
# reset-kernel changes: parity of everything world generation feeds, then the mass time-out step and the headline bench line
out=gpurun_out/${1:-r03_reset}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ymaze.py tests/test_gpu_tmaze.py tests/test_gpu_sim2real.py tests/test_gpu_putnext.py -q -x > $out/tests.log 2>&1
echo "tests rc=$?"; tail -3 $out/tests.log
timeout -k 10 120 python scripts/mass_timeout.py > $out/mass.txt 2>&1; tail -1 $out/mass.txt
timeout -k 10 120 python scripts/mass_timeout.py MiniWorld-FourRooms-v0 > $out/mass_fr.txt 2>&1; tail -1 $out/mass_fr.txt
timeout -k 10 200 python bench.py --steps 300 --warmup 50 --no-vecenv --no-cpu-baseline > $out/bench300.json 2>/dev/null
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-vecenv --no-cpu-baseline > $out/bench20.json 2>/dev/null
python - <<PY
import json
for f in ("bench300","bench20"):
    try:
        d=json.load(open("$out/%s.json"%f)); print(f, round(d["value"]/1e6,3), d["kernel_ms"])
    except Exception as e: print(f, "failed", e)
PY
