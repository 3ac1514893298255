#!/bin/bash
# bench.py --gpus 2 without a launcher: the parent starts its own ranks (two gloo ranks share the one GPU of this box) and relays rank 0's line
mkdir -p gpurun_out
PCACC_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 6 --warmup 6 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model > gpurun_out/r06_bench_selflaunch_2ranks.json 2> gpurun_out/r06_bench_selflaunch_2ranks.err
echo "rc=$?"; tail -1 gpurun_out/r06_bench_selflaunch_2ranks.json | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('n_gpus', 'value', 'ms_per_step', 'scaling')}); print({k: (d.get('config') or {}).get(k) for k in ('ranks_seen', 'per_rank_ms_per_step', 'exposed_allreduce_ms', 'rank_devices', 'backend')})"
