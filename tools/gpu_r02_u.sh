#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_tube.py tests/test_hip_ops.py tests/test_step.py -q -x -p no:cacheprovider 2>&1 | tail -4 | cut -c1-300
timeout 900 python -m pytest tests/test_model_parity.py tests/test_config_parity.py tests/test_cluster.py -q -x -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300
for i in 1 2 3; do timeout 900 python bench.py --no-cpu-baseline --no-fp32-leg 2>&1 | tail -1 | tee gpurun_out/bench_u.json | cut -c100-200; done
