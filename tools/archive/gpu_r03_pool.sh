#!/bin/bash
set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_mlp_split.py tests/test_hip_ops.py -x -q 2>&1 | tail -8
timeout 900 python -m pytest tests/test_config_parity.py -x -q -m gpu -k "c3 and not lidar" 2>&1 | tail -5
timeout 600 python bench.py --steps 10 --warmup 3 --no-step-model --no-cpu-baseline --no-configs 2>&1 | tail -1 | cut -c1-400
