#!/bin/bash
# round 5: chosen kernels INSIDE the step (one-stream steady-state summary under rocprofv3), this tree against the previous commit's library.  usage: ... "<grep pattern>"
PAT=${1:-bilinear}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "" $R/build/libpcacc_hip_prev.so; do
  rm -rf /tmp/prof_ab
  PCACC_LIB=$lib timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o bench -- python3 $R/bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --one-stream > /tmp/rocprof_ab.log 2>&1
  python3 $R/tools/kstats_steady.py /tmp/prof_ab/bench_kernel_trace.csv 6 400 > /tmp/steady_ab.txt
  echo "== lib=${lib:-tree}"; head -3 /tmp/steady_ab.txt | cut -c1-150; grep "$PAT" /tmp/steady_ab.txt | cut -c1-70,100-170
done
