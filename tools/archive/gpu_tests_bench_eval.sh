#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15 > gpurun_out/pytest_gpu.log; tail -15 gpurun_out/pytest_gpu.log
echo "=== bench default"; timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-configs 2>&1 | tail -1 | tee gpurun_out/bench_b4.log | cut -c1-400
echo "=== eval"; timeout 600 python tools/bench_eval.py 2>&1 | tail -4 | tee gpurun_out/bench_eval.log | cut -c1-300
tar czf gpurun_out/miopen_cache.tgz .miopen_cache 2>/dev/null
