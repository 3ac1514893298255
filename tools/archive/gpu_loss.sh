#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/bench_loss.py 2>&1 | grep '^{' | tee gpurun_out/bench_loss.jsonl | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_loss -o loss -- python3 $GRAFT_REPO_ROOT/tools/bench_loss.py --no-host --iters 10 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_loss.log 2>&1
cd $GRAFT_REPO_ROOT; python3 tools/kstats.py gpurun_out/prof_loss/loss_kernel_stats.csv 9 30 > gpurun_out/loss_summary.txt; head -40 gpurun_out/loss_summary.txt | cut -c1-170
