#!/bin/bash
# steady-state kernel summary of the default bench step under rocprofv3 (kernel trace only) -> gpurun_out/${1:-r06_mixed_steady}.txt ; extra bench flags in $PCACC_BENCH_EXTRA
NAME=${1:-r06_mixed_steady}
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$NAME
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$NAME -o bench -- python3 $R/bench.py --steps 6 --warmup 6 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model $PCACC_BENCH_EXTRA > $R/gpurun_out/rocprof_$NAME.log 2>&1
cd $R; tail -1 gpurun_out/rocprof_$NAME.log | cut -c1-200
T=$(find gpurun_out/prof_$NAME -name "*kernel_trace.csv" | head -1)
PCACC_KSTATS_DUMP=gpurun_out/${NAME}_launches.txt python3 tools/kstats_steady.py $T 6 400 > gpurun_out/$NAME.txt
head -12 gpurun_out/$NAME.txt | cut -c1-170
S=$(find gpurun_out/prof_$NAME -name "*kernel_stats.csv" | head -1)
cp $S gpurun_out/${NAME}_kernel_stats.csv
rm -rf gpurun_out/prof_$NAME
