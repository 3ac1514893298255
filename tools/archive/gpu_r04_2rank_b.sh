#!/bin/bash
# which round-4 change hangs the two-ranks-on-one-GPU bench?  (each run: 100 s watchdog dumping all Python stacks)
mkdir -p gpurun_out
run() { echo "== $1 | $2"; env $1 PCACC_HANG_DUMP=100 PCACC_DIST_BACKEND=gloo timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) bench.py --gpus 2 --steps 3 --warmup 2 --batch 2 --no-cpu-baseline --no-fp32-leg $2 > gpurun_out/2rank_$3.txt 2>&1; tail -1 gpurun_out/2rank_$3.txt | cut -c1-150; grep -n "File \"/root/repo\|Thread\|most recent" gpurun_out/2rank_$3.txt | head -40; }
run "PCACC_X=0" "" default
run "PCACC_SPARSE_EGO=0" "" nosparse
run "PCACC_BATCHED_COLLATE=0" "" nobatch
run "PCACC_X=0" "--dtype bf16" bf16
run "PCACC_TWO_STREAMS_DIST=0" "" plain
