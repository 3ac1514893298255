#!/bin/bash
# steady-state kernel stats (one stream) of the bf16 and fp32x3 steps + call tables
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for d in bf16 fp32x3; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$d -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --dtype $d --steps 5 --warmup 3 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --one-stream > $GRAFT_REPO_ROOT/gpurun_out/rocprof_$d.log 2>&1
done
cd $GRAFT_REPO_ROOT
for d in bf16 fp32x3; do
  python3 tools/kstats_steady.py gpurun_out/prof_$d/bench_kernel_trace.csv 4 200 > gpurun_out/${d}_steady.txt; head -9 gpurun_out/${d}_steady.txt
  PCACC_DTYPE=$d timeout 600 python tools/native_call_table.py 20 > gpurun_out/call_table_$d.txt 2>&1
done
grep -i "igemm\|ck::\|_ZN2ck\|miopen\|naive_conv" gpurun_out/fp32x3_steady.txt | cut -c1-150 | head
