#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_ops.py -q -x -p no:cacheprovider -k "pfn_block or pfn" 2>&1 | tail -15 | cut -c1-300
echo "=== bench"; timeout 900 python bench.py --no-cpu-baseline --no-fp32-leg 2>&1 | tail -1 | tee gpurun_out/bench_pfn.json | cut -c1-200
timeout 600 python tools/native_call_table.py 60 2>&1 | grep -v Warn > gpurun_out/native_calls.txt; grep -n "pfn_block" gpurun_out/native_calls.txt | cut -c1-200
