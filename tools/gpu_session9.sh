#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/pytest_gpu.log; tail -2 gpurun_out/pytest_gpu.log
echo "=== bench default"; timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_b4.log | cut -c1-260
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1
cd $GRAFT_REPO_ROOT; tar czf gpurun_out/miopen_cache.tgz .miopen_cache 2>/dev/null
