#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/pytest_gpu.log; tail -2 gpurun_out/pytest_gpu.log
for b in 1 4; do echo "=== bench batch $b"; timeout 600 python bench.py --steps 8 --warmup 3 --batch $b --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_b$b.log | cut -c1-260; done
timeout 600 python tools/host_profile.py 2>&1 | grep -v -E "Warning|detach|amdgpu" > gpurun_out/host_profile.log; head -3 gpurun_out/host_profile.log
tar czf gpurun_out/miopen_cache.tgz .miopen_cache 2>/dev/null
