#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_tube.py -q -x -p no:cacheprovider 2>&1 | tail -25 | cut -c1-250
timeout 900 python -m pytest tests/test_model_parity.py tests/test_config_parity.py -q -x -p no:cacheprovider 2>&1 | tail -5 | cut -c1-300
echo "=== bench"; timeout 900 python bench.py --no-cpu-baseline --no-fp32-leg 2>&1 | tail -1 | tee gpurun_out/bench_tube.json | cut -c1-200
timeout 600 python tools/stage_backward_times.py 2>&1 | grep -v Warning | tail -14 | tee gpurun_out/stage_backward_times.txt
