#!/bin/bash
# One GPU-box session: GPU test suite, the headline bench line, rocprofv3 kernel stats of the same
# command.  Usage (from the repo root, through gpurun):  bash tools/gpu_session.sh <tag> [what...]
#   what: tests bench bench1 prof pmc benchq dist1 cli cold soak fuzz recipes driver  (default: tests bench prof)
set -u
TAG=${1:-s}; shift || true
WHAT=${*:-tests bench prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then
  timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log
fi
if has bench; then
  timeout 1200 python bench.py --steps 3 --warmup 1 > $OUT/bench_256.json 2> $OUT/bench_256.err; echo "bench rc=$?"
  tail -c 3000 $OUT/bench_256.json
fi
if has bench1; then
  timeout 600 python bench.py --steps 10 --warmup 3 --clips-per-gpu 1 --no-cpu-baseline > $OUT/bench_1clip.json 2> $OUT/bench_1clip.err; echo "bench1 rc=$?"
  timeout 900 python bench.py --steps 2 --warmup 1 --kind separator --no-cpu-baseline > $OUT/bench_separator_256.json 2> $OUT/bench_separator.err; echo "benchsep rc=$?"
fi
if has prof; then
  (cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof_bench.err); echo "rocprof rc=$?"
  find /tmp/prof_kt -name "*kernel_stats.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_bench_256clips.csv \;
  head -20 $OUT/rocprofv3_kernel_stats_bench_256clips.csv
fi
if has pmc; then
  for pass in "A SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "C FETCH_SIZE" "D WRITE_SIZE"; do
    set -- $pass; name=$1; shift
    (cd /tmp && timeout 2400 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-pass > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/pmc_$name.err); echo "pmc $name rc=$?"
  done
  python tools/pmc_summary.py $OUT/pmc_summary_bench_256clips.json /tmp/pmc_A /tmp/pmc_C /tmp/pmc_D && head -c 1500 $OUT/pmc_summary_bench_256clips.json
fi
if has benchq; then
  # the headline line once more, now that the PMC summary of THESE kernel sources exists: roofline.traffic is quoted
  mkdir -p profiles/$TAG && cp $OUT/pmc_summary_bench_256clips.json profiles/$TAG/pmc_summary_bench_256clips.json
  timeout 1200 python bench.py --steps 3 --warmup 1 > $OUT/bench_256_with_traffic.json 2> $OUT/bench_256_with_traffic.err; echo "benchq rc=$?"
  tail -c 1200 $OUT/bench_256_with_traffic.json
fi
if has dist1; then
  # the world > 1 branch of bench.py on the one GPU there is: 2 ranks share device 0, gloo all-gather
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 2 --steps 2 --warmup 1 --clips-per-gpu 8 --share-device0 --no-kernel-pass > $OUT/bench_2ranks_shared.json 2> $OUT/bench_2ranks_shared.err; echo "dist1 rc=$?"
  tail -c 1500 $OUT/bench_2ranks_shared.json
  # and its RCCL leg with the one rank a one-GPU box allows (communicator init, all-gather, barriers, all-reduce)
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
      bench.py --gpus 1 --steps 2 --warmup 1 --clips-per-gpu 16 --force-dist --no-cpu-baseline > $OUT/bench_rccl_1rank.json 2> $OUT/bench_rccl_1rank.err; echo "rccl1 rc=$?"
fi
if has cli; then      # files in -> files out through the command line, 256 wavs (first call and steady state)
  timeout 900 python tools/cli_e2e.py 256 10 > $OUT/cli_e2e_256clips.log 2>&1; tail -c 1200 $OUT/cli_e2e_256clips.log
fi
if has cold; then     # one 10 s file per fresh process: configs[1] as a user meets it
  timeout 600 python tools/cold_call.py 3 10 > $OUT/cold_call.json 2>$OUT/cold_call.err; grep -A6 summary_wall_s $OUT/cold_call.json
fi
if has soak; then
  timeout 900 python tools/soak.py 60 4 > $OUT/soak.log 2>&1; tail -3 $OUT/soak.log
fi
if has fuzz; then
  timeout 1200 python tools/fuzz_batches.py 150 > $OUT/fuzz_batches.log 2>&1; tail -3 $OUT/fuzz_batches.log
fi
if has recipes; then  # the parity tables with their absolute errors printed
  timeout 900 python -m pytest tests/test_gpu_recipes.py tests/test_gpu_full10s.py -m gpu -q -s > $OUT/pytest_gpu_recipes_full10s.log 2>&1; tail -2 $OUT/pytest_gpu_recipes_full10s.log
fi
if has driver; then   # the driver's own command
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_256clips_driver_shape_20steps.json 2>$OUT/bench_driver_shape.err; tail -c 600 $OUT/bench_256clips_driver_shape_20steps.json
fi
