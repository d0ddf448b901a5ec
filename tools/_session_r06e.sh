mkdir -p gpurun_out/r06e
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -m gpu > gpurun_out/r06e/pytest.log 2>&1; tail -3 gpurun_out/r06e/pytest.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r06e/bench_new_direct.json 2>gpurun_out/r06e/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r06e/bench_new_direct.json')); print('new', round(d['value'],1), round(d['ms_per_step'],1), {k: round(v,1) for k, v in d['kernel_ms_per_step'].items() if v > 20})"
bash tools/ab_variant_libs.sh 2 > gpurun_out/r06e/ab_pipe.txt 2>&1; cat gpurun_out/r06e/ab_pipe.txt
