mkdir -p gpurun_out/r06g
timeout 900 python tools/ab_variants.py --option transform_kernel --variants 2 0 1 --clips 64 --rounds 3 > gpurun_out/r06g/ab_transform_kernel.txt 2>&1; grep -v "^  [a-z_]*  \|avgpool\|cond_proj\|stft\|istft" gpurun_out/r06g/ab_transform_kernel.txt | grep "variant\|total\|dma<128>\|conv_igemm<" 
NHANS_LIB=$PWD/build_ab/libnhans_hip_dev.so timeout 600 python tools/wino_phase_cycles.py 1 1700 > gpurun_out/r06g/wino_phase_cycles_trimmed_sweep.txt 2>&1; grep "wave 0\|wave 4\|block" gpurun_out/r06g/wino_phase_cycles_trimmed_sweep.txt | cut -c1-420
