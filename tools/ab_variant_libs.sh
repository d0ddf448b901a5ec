#!/bin/bash
# Same-box A/B of the libraries under build_ab/variants (tools/build_variant.sh) -- or under $VARIANTS_DIR: the default bench
# alternately with each, ROUNDS times; prints x real-time, ms per step and the conv kernels' ms per step, after one
# tools/lib_fingerprint.py line per library (builds that are meant to differ in speed only must agree bit for bit).
# Boxes differ by several per cent: only a same-box A/B is honest.
#     gpurun -- 'bash tools/ab_variant_libs.sh [ROUNDS] [bench flags]'
rounds=${1:-2}; shift
dir=${VARIANTS_DIR:-build_ab/variants}
mkdir -p gpurun_out/ab
for lib in $dir/libnhans_*.so; do
  NHANS_LIB=$PWD/$lib python tools/lib_fingerprint.py 2>/dev/null | tr '\n' ' '; echo
done
for r in $(seq $rounds); do
for lib in $dir/libnhans_*.so; do
  n=$(basename $lib .so); n=${n#libnhans_}
  NHANS_LIB=$PWD/$lib python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ceiling "$@" > gpurun_out/ab/$n.$r.json 2>gpurun_out/ab/$n.$r.err
  python -c "
import json; d=json.load(open('gpurun_out/ab/$n.$r.json')); print('%-12s' % '$n', round(d['value'],1), round(d['ms_per_step'],1), {k: round(v,1) for k, v in d['kernel_ms_per_step'].items() if k.startswith('conv_') and v > 100})"
done; done
