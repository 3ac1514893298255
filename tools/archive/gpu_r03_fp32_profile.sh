#!/bin/bash
# round 3 starting point: kernel-time breakdown of the fp32 (matched-accuracy) step with library convolutions
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_fp32 -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --dtype fp32 --steps 4 --warmup 2 --no-cpu-baseline --no-configs --no-miopen-find --one-stream > $GRAFT_REPO_ROOT/gpurun_out/rocprof_fp32.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/rocprof_fp32.log | cut -c1-300
python3 tools/kstats.py gpurun_out/prof_fp32/bench_kernel_stats.csv 6 70 > gpurun_out/fp32_summary.txt; head -12 gpurun_out/fp32_summary.txt | cut -c1-170
