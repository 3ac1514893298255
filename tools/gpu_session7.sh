#!/bin/bash
mkdir -p gpurun_out
echo "=== bench default (batch 4) with cpu baseline"
timeout 1500 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_full.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/rocprof_bench.log | cut -c1-200
tar czf gpurun_out/miopen_cache.tgz .miopen_cache 2>/dev/null
