#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_conv.py -q -x -p no:cacheprovider 2>&1 | tail -6 | cut -c1-300
echo "=== bench"; timeout 900 python bench.py --no-cpu-baseline --no-fp32-leg 2>&1 | tail -1 | tee gpurun_out/bench_ms.json | cut -c1-200
