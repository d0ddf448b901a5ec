mkdir -p gpurun_out/r06f
VARIANTS_DIR=build_ab/variants_wl bash tools/ab_variant_libs.sh 2 > gpurun_out/r06f/ab_wholeline.txt 2>&1; cat gpurun_out/r06f/ab_wholeline.txt
VARIANTS_DIR=build_ab/variants_trim bash tools/ab_variant_libs.sh 3 > gpurun_out/r06f/ab_trim.txt 2>&1; cat gpurun_out/r06f/ab_trim.txt
