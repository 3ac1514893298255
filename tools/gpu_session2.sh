#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "model_parity or test_gpu" 2>&1 | tail -80 > gpurun_out/pytest_model.log
tail -40 gpurun_out/pytest_model.log
echo "=== bench bf16"
timeout 900 python bench.py --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/bench_bf16.log 2>&1; tail -3 gpurun_out/bench_bf16.log
echo "=== bench fp32"
timeout 900 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --dtype fp32 > gpurun_out/bench_fp32.log 2>&1; tail -3 gpurun_out/bench_fp32.log
echo "=== bench bf16 NHWC hint"
PYTORCH_MIOPEN_SUGGEST_NHWC=1 timeout 900 python bench.py --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/bench_bf16_nhwc.log 2>&1; tail -3 gpurun_out/bench_bf16_nhwc.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/prof_bench | head
