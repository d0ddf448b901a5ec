#!/bin/bash
# Same-box A/B of two BUILDS of the library (a kernel change, not an option): build the baseline, copy it to
# n-hans_amd/csrc/libnhans_hip_base.so (git-ignored, travels with gpurun), rebuild the candidate, then
#     gpurun -- 'bash tools/ab_libs.sh'
# runs the default bench alternately with both ($NHANS_LIB selects the library in nhans_amd/hip.py) and prints
# x real-time, ms per step and the Winograd kernel's ms per step.  Boxes differ by several per cent; only this is honest.
for r in 1 2; do
for lib in base new; do
  if [ $lib = base ]; then export NHANS_LIB=$PWD/n-hans_amd/csrc/libnhans_hip_base.so; else unset NHANS_LIB; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ceiling > gpurun_out/ab_$lib.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/ab_$lib.json')); print('$lib', round(d['value'],1), round(d['ms_per_step'],1), {k: round(v,1) for k, v in d['kernel_ms_per_step'].items() if k.startswith('conv_')})"
done; done
