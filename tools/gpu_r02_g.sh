#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/bf16_deltas.jsonl
R=$GRAFT_REPO_ROOT
PCACC_DUMP_DELTAS=1 timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/pytest_gpu.log; tail -8 gpurun_out/pytest_gpu.log | cut -c1-300
grep trained gpurun_out/bf16_deltas.jsonl | cut -c1-600
echo "=== bench default"; timeout 1500 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.json; cut -c1-2600 gpurun_out/bench_default.json
echo "=== 2 ranks (gloo, both on cuda:0)"; PCACC_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 2 2>&1 | tail -2 | cut -c1-500 | tee gpurun_out/bench_2rank_gloo.log
