mkdir -p gpurun_out/r06t
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream_kernel" 2>&1 | tail -3
timeout 900 python tools/ab_variants.py --option stream_1x1 --variants 0 1 --clips 64 --rounds 3 > gpurun_out/r06t/ab_stream_1x1.txt 2>&1; grep "variant\|total\|dma<128>\|1x1" gpurun_out/r06t/ab_stream_1x1.txt
bash tools/gpu_session.sh r06t tests bench prof pmc benchq driver bench1 soak fuzz
