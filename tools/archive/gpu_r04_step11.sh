#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_mixed.py tests/test_train_trajectory.py -q -m gpu -k "bn or batch_norm or mixed or sparse or trajectory or stale" > gpurun_out/t_part.txt 2>&1; tail -5 gpurun_out/t_part.txt
timeout 1500 python -m pytest tests/test_config_parity.py tests/test_model_parity.py tests/test_step.py -q -m gpu > gpurun_out/t_part2.txt 2>&1; tail -5 gpurun_out/t_part2.txt
for i in 1 2 3; do
  ms=$(timeout 900 python bench.py --dtype mixed --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>gpurun_out/err_mixed.txt | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "mixed B=4 $ms"
done
cd /tmp && export TMPDIR=/tmp
for d in mixed; do
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$d -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --dtype $d --steps 5 --warmup 3 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --one-stream > $GRAFT_REPO_ROOT/gpurun_out/rocprof_$d.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kstats_steady.py $GRAFT_REPO_ROOT/gpurun_out/prof_$d/bench_kernel_trace.csv 4 200 > $GRAFT_REPO_ROOT/gpurun_out/r04_${d}_steady_v8.txt; head -9 $GRAFT_REPO_ROOT/gpurun_out/r04_${d}_steady_v8.txt | cut -c1-190
done
grep -n "absmax256\|bn_apply\|bfloat16_copy" $GRAFT_REPO_ROOT/gpurun_out/r04_mixed_steady_v8.txt | cut -c1-220
