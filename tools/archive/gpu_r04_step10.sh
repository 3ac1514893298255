#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_upconv_bf16.py -q -m gpu > gpurun_out/t_up.txt 2>&1; tail -3 gpurun_out/t_up.txt
cd /tmp && export TMPDIR=/tmp
for d in mixed; do
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$d -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --dtype $d --steps 5 --warmup 3 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --one-stream > $GRAFT_REPO_ROOT/gpurun_out/rocprof_$d.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kstats_steady.py $GRAFT_REPO_ROOT/gpurun_out/prof_$d/bench_kernel_trace.csv 4 200 > $GRAFT_REPO_ROOT/gpurun_out/r04_${d}_steady_v7.txt; head -9 $GRAFT_REPO_ROOT/gpurun_out/r04_${d}_steady_v7.txt | cut -c1-190
done
grep -n "upconv_bf16" $GRAFT_REPO_ROOT/gpurun_out/r04_mixed_steady_v7.txt | cut -c1-220
