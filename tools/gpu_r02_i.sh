#!/bin/bash
# early backward (DataParallelStep.pipelined) on / off, same box
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_parity.py tests/test_config_parity.py -q -x -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300
for i in 1 2; do
echo "=== bench pipelined"; timeout 900 python bench.py --no-cpu-baseline --no-fp32-leg 2>&1 | tail -1 | tee gpurun_out/bench_pipe_$i.json | cut -c1-200
echo "=== bench single backward"; timeout 900 python bench.py --no-cpu-baseline --no-fp32-leg --no-pipeline 2>&1 | tail -1 | tee gpurun_out/bench_nopipe_$i.json | cut -c1-200
done
