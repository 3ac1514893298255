#!/bin/bash
# final tree: default bench as the driver runs it; the same command under rocprofv3 --kernel-trace --stats (kernel stats csv)
mkdir -p gpurun_out
bash tools/gpu_default_bench.sh; cp gpurun_out/bench_default_full.json gpurun_out/r06_bench_default_full.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_default
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_default -o bench -- python3 $R/bench.py --no-cpu-baseline --no-configs > $R/gpurun_out/rocprof_default.log 2>&1
cp $(find $R/gpurun_out/prof_default -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r06_default_kernel_stats.csv
grep -n "seg_max_canvas" $R/gpurun_out/r06_default_kernel_stats.csv | cut -c1-220
grep "^{" $R/gpurun_out/rocprof_default.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('under rocprofv3:', round(d['ms_per_step'],2), 'roofline kernel avg us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
rm -rf $R/gpurun_out/prof_default
