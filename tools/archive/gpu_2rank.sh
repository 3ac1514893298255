#!/bin/bash
# two gloo ranks sharing cuda:0 (the only N > 1 configuration a 1-GPU box can run): tools/gpu_2rank.sh "<extra bench flags>"
PCACC_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) bench.py --gpus 2 --steps 4 --warmup 2 $1 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],1))" 2>&1 | tail -1
