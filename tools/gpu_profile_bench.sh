#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-configs --no-fp32-leg --no-miopen-find $PCACC_BENCH_EXTRA > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.log 2>&1
cd $GRAFT_REPO_ROOT; tail -1 gpurun_out/rocprof_bench.log | cut -c1-200
python3 tools/kstats.py gpurun_out/prof_bench/bench_kernel_stats.csv 9 40 > gpurun_out/bench_summary.txt; head -50 gpurun_out/bench_summary.txt | cut -c1-170
python3 tools/trace_timeline.py gpurun_out/prof_bench/bench_kernel_trace.csv 6 | tee gpurun_out/bench_timeline.txt
