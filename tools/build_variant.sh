#!/bin/bash
# Dev tool: a variant of the library that differs from the built one in conv_wino.hip's compile flags only:
#     tools/build_variant.sh NAME -DFLAG=... -> bin_tmp/variants/libnhans_NAME.so   (run with tools/ab_variant_libs.sh on the GPU box)
set -e
name=$1; shift
cd /root/repo/n-hans_amd/csrc
make -s >/dev/null
mkdir -p /root/repo/bin_tmp/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize "$@" -c conv_wino.hip -o /root/repo/bin_tmp/variants/conv_wino_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/bin_tmp/variants/libnhans_$name.so /root/repo/bin_tmp/variants/conv_wino_$name.o \
  $(for f in conv_igemm conv_igemm_dma conv_igemm_halo aux_kernels stft mfma_ceiling launch_status nhans_api; do echo $f.o; done)
ls -la /root/repo/bin_tmp/variants/libnhans_$name.so
