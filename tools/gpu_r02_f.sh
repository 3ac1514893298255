#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/bf16_deltas.jsonl
R=$GRAFT_REPO_ROOT
PCACC_DUMP_DELTAS=1 timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -30 > gpurun_out/pytest_gpu.log; tail -12 gpurun_out/pytest_gpu.log | cut -c1-300
echo "=== bench_conv"; timeout 600 python tools/bench_conv.py 2>&1 | grep '^{' | tee gpurun_out/bench_conv.jsonl | grep -E "128|256|512" | cut -c1-260
echo "=== bench (no cpu baseline)"; timeout 900 python bench.py --no-cpu-baseline --no-fp32-leg 2>&1 | tail -1 | tee gpurun_out/bench_nocpu.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-miopen-find > $R/gpurun_out/rocprof_bench.log 2>&1
cd $R
python3 tools/kstats.py gpurun_out/prof_bench/bench_kernel_stats.csv 9 110 > gpurun_out/bench_summary.txt; head -40 gpurun_out/bench_summary.txt | cut -c1-170
