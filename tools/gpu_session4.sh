#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
echo "=== stage times 160k bf16"
timeout 600 python tools/stage_times.py 160000 bf16 3 2>&1 | grep -v -E "amdgpu.ids|Warning|detach|stats =" | tee gpurun_out/stage_bf16.log | tail -20
echo "=== bench bf16"
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_bf16.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
tar czf gpurun_out/miopen_cache.tgz .miopen_cache 2>/dev/null; ls -la gpurun_out/miopen_cache.tgz
