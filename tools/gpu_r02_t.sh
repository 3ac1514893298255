#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_step.py -q -x -p no:cacheprovider 2>&1 | tail -12 | cut -c1-300
timeout 900 python -m pytest tests/test_model_parity.py tests/test_config_parity.py tests/test_cluster.py -q -x -p no:cacheprovider 2>&1 | tail -4 | cut -c1-300
for i in 1 2; do timeout 900 python bench.py --no-cpu-baseline --no-fp32-leg 2>&1 | tail -1 | tee gpurun_out/bench_2s.json | cut -c100-200; done
