#!/bin/bash
# fp32x3 step: tests of the split kernels, bench line, rocprofv3 kernel stats (one stream: per-kernel times are not inflated by a second queue)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mlp_split.py tests/test_conv_split.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --dtype fp32x3 --steps 8 --warmup 3 --no-cpu-baseline --no-configs 2>&1 | tail -1 > gpurun_out/bench_x3.json; cut -c1-250 gpurun_out/bench_x3.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_x3 -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --dtype fp32x3 --steps 4 --warmup 2 --no-cpu-baseline --no-configs --one-stream > $GRAFT_REPO_ROOT/gpurun_out/rocprof_x3.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/kstats.py gpurun_out/prof_x3/bench_kernel_stats.csv 6 90 > gpurun_out/x3_summary.txt; head -70 gpurun_out/x3_summary.txt | cut -c1-170
