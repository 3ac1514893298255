#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv.py -q -x -p no:cacheprovider 2>&1 | tail -4 | cut -c1-300
echo "=== bench"; timeout 900 python bench.py --no-cpu-baseline --no-fp32-leg 2>&1 | tail -1 | tee gpurun_out/bench_wg.json | cut -c1-200
timeout 600 python tools/native_call_table.py 40 2>&1 | grep -v Warn > gpurun_out/native_calls.txt; grep -n "conv3x3_wgrad " gpurun_out/native_calls.txt | cut -c1-200
