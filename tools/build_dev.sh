#!/bin/bash
# builds the DEV=1 library next to the default one without disturbing it: bin_tmp/libnhans_hip_dev.so
set -e
cd /root/repo/n-hans_amd/csrc
mkdir -p /root/repo/bin_tmp/dev
for f in conv_igemm conv_igemm_dma conv_igemm_halo conv_wino aux_kernels stft mfma_ceiling launch_status nhans_api; do
  if [ ! -f /root/repo/bin_tmp/dev/$f.o ] || [ $f.hip -nt /root/repo/bin_tmp/dev/$f.o ] || [ conv_wino_common.h -nt /root/repo/bin_tmp/dev/$f.o ] || [ nhans_kernels.h -nt /root/repo/bin_tmp/dev/$f.o ] || [ conv_epilogue.h -nt /root/repo/bin_tmp/dev/$f.o ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DNHANS_DEV $( [ $f = conv_wino ] && echo -fno-slp-vectorize ) -c $f.hip -o /root/repo/bin_tmp/dev/$f.o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/bin_tmp/libnhans_hip_dev.so $(for f in conv_igemm conv_igemm_dma conv_igemm_halo conv_wino aux_kernels stft mfma_ceiling launch_status nhans_api; do echo /root/repo/bin_tmp/dev/$f.o; done)
ls -la /root/repo/bin_tmp/libnhans_hip_dev.so
