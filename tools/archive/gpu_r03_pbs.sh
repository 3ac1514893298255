#!/bin/bash
# fused fp32x3 PFN block: unit tests, then the config parity in fp32x3, then the bench leg
set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_mlp_split.py -x -q 2>&1 | tail -15 > gpurun_out/pbs_tests.log
tail -5 gpurun_out/pbs_tests.log
timeout 900 python -m pytest tests/test_config_parity.py -x -q -m gpu -k "fp32x3" 2>&1 | tail -15 > gpurun_out/pbs_parity.log
tail -5 gpurun_out/pbs_parity.log
timeout 600 python bench.py --dtype fp32x3 --steps 10 --warmup 3 --no-step-model 2>&1 | tail -3 > gpurun_out/pbs_bench.log
tail -2 gpurun_out/pbs_bench.log
