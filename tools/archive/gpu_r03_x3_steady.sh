#!/bin/bash
# steady-state kernel stats (one stream) + native call table of the fp32x3 step
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_fp32x3 -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --dtype fp32x3 --steps 5 --warmup 3 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --one-stream > $GRAFT_REPO_ROOT/gpurun_out/rocprof_fp32x3.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/kstats_steady.py gpurun_out/prof_fp32x3/bench_kernel_trace.csv 4 200 > gpurun_out/fp32x3_steady.txt; head -60 gpurun_out/fp32x3_steady.txt | cut -c1-180
PCACC_DTYPE=fp32x3 timeout 600 python tools/native_call_table.py 20 > gpurun_out/call_table_fp32x3.txt 2>&1
head -50 gpurun_out/call_table_fp32x3.txt | cut -c1-160
