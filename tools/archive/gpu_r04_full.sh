#!/bin/bash
# the whole GPU suite (no -x: every failure listed), then the default bench command
mkdir -p gpurun_out
t0=$(date +%s)
timeout 4000 python -m pytest tests -q -m gpu > gpurun_out/t_full.txt 2>&1; tail -15 gpurun_out/t_full.txt
echo "suite wall $(( $(date +%s) - t0 )) s"
bash tools/gpu_default_bench.sh
