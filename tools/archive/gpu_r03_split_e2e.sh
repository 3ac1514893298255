#!/bin/bash
# fp32x3 mode end to end: whole-model parity at config size, step time, kernel profile
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_config_parity.py -x -q -k "fp32x3" 2>&1 | tail -15
timeout 600 python bench.py --dtype fp32x3 --steps 6 --warmup 3 --no-cpu-baseline --no-configs 2>&1 | tail -1 | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_x3 -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --dtype fp32x3 --steps 4 --warmup 2 --no-cpu-baseline --no-configs --one-stream > $GRAFT_REPO_ROOT/gpurun_out/rocprof_x3.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/kstats.py gpurun_out/prof_x3/bench_kernel_stats.csv 6 80 > gpurun_out/x3_summary.txt; head -12 gpurun_out/x3_summary.txt | cut -c1-170
