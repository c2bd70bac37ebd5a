for rep in 1 2; do
for v in "MWB_NO_FUSED_PREP=1" "MWB_X=1"; do
  env $v python bench.py --no-cpu-baseline --no-vecenv --workload maze8192 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,3), round(d['ms_per_step'],4), d['kernel_ms'])"
done; done
