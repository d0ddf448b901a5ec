mkdir -p gpurun_out/r06o
for r in 1 2; do
for lib in build_ab/variants_d4/libnhans_*.so; do
  n=$(basename $lib .so); n=${n#libnhans_}
  NHANS_LIB=$PWD/$lib python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ceiling --clips-per-gpu 128 > gpurun_out/r06o/$n.$r.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r06o/$n.$r.json')); print('%-10s' % '$n', round(d['value'],1), {k: round(v,2) for k, v in d['kernel_ms_per_step'].items() if k.startswith('direct') or k.startswith('conv_wino')})"
done; done > gpurun_out/r06o/ab_direct_conv64_grid_and_store_policy.txt 2>&1
cat gpurun_out/r06o/ab_direct_conv64_grid_and_store_policy.txt
