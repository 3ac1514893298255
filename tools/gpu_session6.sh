#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
echo "=== bench bf16"
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_bf16.log | cut -c1-330
echo "=== 2 ranks on one GPU over gloo (N>1 code path)"
PCACC_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 2>&1 | tail -2 | tee gpurun_out/bench_2rank_gloo.log | cut -c1-330
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
tar czf gpurun_out/miopen_cache.tgz .miopen_cache 2>/dev/null
