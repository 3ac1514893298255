#!/bin/bash
# end-of-round evidence on the final tree: per-config table, LiDAR-shaped bench, counter passes, call tables, steady profiles (mixed, bf16)
mkdir -p gpurun_out
timeout 1500 python tools/bench_configs.py 8 2>gpurun_out/err_configs.txt | grep "^{" > gpurun_out/r04_bench_configs.jsonl; cut -c1-200 gpurun_out/r04_bench_configs.jsonl
timeout 900 python bench.py --points lidar --no-configs > gpurun_out/r04_bench_lidar.json 2>gpurun_out/err_lidar.txt; tail -1 gpurun_out/r04_bench_lidar.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lidar', d['ms_per_step'], d.get('bf16',{}).get('ms_per_step'))"
for d in mixed bf16; do
  PCACC_DTYPE=$d timeout 600 python tools/native_call_table.py 20 > gpurun_out/r04_native_call_table_${d}_final.txt 2>&1; head -4 gpurun_out/r04_native_call_table_${d}_final.txt | cut -c1-160
done
bash tools/gpu_pmc_kernels.sh > gpurun_out/pmc_kernels.log 2>&1; tail -3 gpurun_out/pmc_kernels.log | cut -c1-200
cp gpurun_out/pmc_kernels_summary.json gpurun_out/r04_pmc_kernels_summary.json
cd /tmp && export TMPDIR=/tmp
for d in mixed bf16; do
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$d -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --dtype $d --steps 5 --warmup 3 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --one-stream > $GRAFT_REPO_ROOT/gpurun_out/rocprof_$d.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kstats_steady.py $GRAFT_REPO_ROOT/gpurun_out/prof_$d/bench_kernel_trace.csv 4 200 > $GRAFT_REPO_ROOT/gpurun_out/r04_${d}_steady_final.txt; head -9 $GRAFT_REPO_ROOT/gpurun_out/r04_${d}_steady_final.txt | cut -c1-190
done
# the default command under the profiler: the summary the roofline object's kernel duration must agree with
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_default -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-configs > $GRAFT_REPO_ROOT/gpurun_out/rocprof_default.log 2>&1
head -30 $GRAFT_REPO_ROOT/gpurun_out/prof_default/bench_kernel_stats.csv > $GRAFT_REPO_ROOT/gpurun_out/r04_default_kernel_stats_head.csv
grep -n "pillar_scatter" $GRAFT_REPO_ROOT/gpurun_out/prof_default/bench_kernel_stats.csv | cut -c1-200
tail -1 $GRAFT_REPO_ROOT/gpurun_out/rocprof_default.log | cut -c1-300
