#!/bin/bash
# default bench line + rocprofv3 kernel stats of the same workload (heuristic-mode convs so the find kernels do not pollute the totals)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
echo "=== bench default"; timeout 1200 python bench.py 2>&1 | tail -3 > gpurun_out/bench_default.json; cut -c1-2500 gpurun_out/bench_default.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-miopen-find > $R/gpurun_out/rocprof_bench.log 2>&1
cd $R
tail -2 gpurun_out/rocprof_bench.log | cut -c1-600
python3 tools/kstats.py gpurun_out/prof_bench/bench_kernel_stats.csv 9 90 > gpurun_out/bench_summary.txt; head -100 gpurun_out/bench_summary.txt | cut -c1-170
