#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_tube.py -q -x -p no:cacheprovider 2>&1 | tail -12 | cut -c1-250
echo "=== bench"; timeout 900 python bench.py --no-cpu-baseline --no-fp32-leg 2>&1 | tail -1 | tee gpurun_out/bench_few.json | cut -c1-200
timeout 600 python tools/native_call_table.py 60 2>&1 | grep -v Warn > gpurun_out/native_calls.txt; grep -n "csr_build\|floa(3200000, 9)\|floa(422467, 2)\|floa(320000, 4)\|floa(422467, 3)" gpurun_out/native_calls.txt | cut -c1-200
