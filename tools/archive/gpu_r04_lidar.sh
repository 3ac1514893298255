#!/bin/bash
# round 4, verdict item 1(a): the fp32x3 step on LiDAR-shaped scenes (73.6 ms in profiles/r03_bench_lidar.json vs 44 uniform) -- reproduce,
# per-call table for the mode x point-distribution pair, steady kernel stats of the lidar fp32x3 step
mkdir -p gpurun_out
for i in 1 2; do
for p in uniform lidar; do
  ms=$(timeout 900 python bench.py --dtype fp32x3 --points $p --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>gpurun_out/err_$p.txt | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "fp32x3 $p $ms"
done
done
for p in uniform lidar; do
  PCACC_DTYPE=fp32x3 PCACC_POINTS=$p timeout 600 python tools/native_call_table.py 20 > gpurun_out/r04_native_call_table_fp32x3_$p.txt 2>&1
  head -30 gpurun_out/r04_native_call_table_fp32x3_$p.txt | cut -c1-200
done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_x3_lidar -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --dtype fp32x3 --points lidar --steps 5 --warmup 3 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --one-stream > $GRAFT_REPO_ROOT/gpurun_out/rocprof_x3_lidar.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/kstats_steady.py gpurun_out/prof_x3_lidar/bench_kernel_trace.csv 4 200 > gpurun_out/r04_fp32x3_lidar_steady.txt; head -40 gpurun_out/r04_fp32x3_lidar_steady.txt | cut -c1-200
