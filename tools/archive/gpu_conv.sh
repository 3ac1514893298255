#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -25 > gpurun_out/pytest_conv.log; tail -25 gpurun_out/pytest_conv.log
echo "=== bench_conv"; timeout 900 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_conv.log
tar czf gpurun_out/miopen_cache.tgz .miopen_cache 2>/dev/null
