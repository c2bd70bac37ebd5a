for pad in 0 1 2 4 6 7 8 16; do
  MWB_DEBUG=$((pad*256)) python bench.py --no-cpu-baseline --no-vecenv --workload maze8192 --steps 100 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pad', $pad*128, round(d['value']/1e6,3), round(d['kernel_ms']['render'],4))"
done
