#!/bin/bash
# batched weight preparation: tests, A/B of the step (mixed + bf16, B = 4 and c3 B = 1), steady profile
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_prepare_batch.py tests/test_mixed.py tests/test_train_trajectory.py tests/test_conv_split.py tests/test_conv.py tests/test_step.py -q -m gpu > gpurun_out/t_part.txt 2>&1; tail -8 gpurun_out/t_part.txt
for i in 1 2; do for e in "PCACC_BATCH_PREPARE=0" "PCACC_BATCH_PREPARE=1"; do for d in mixed bf16; do
  ms=$(env $e timeout 900 python bench.py --dtype $d --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>gpurun_out/err_mixed.txt | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$e $d B=4 $ms"
  ms=$(env $e timeout 900 python bench.py --dtype $d --batch 1 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>gpurun_out/err_mixed.txt | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$e $d B=1 $ms"
done; done; done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mixed -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --dtype mixed --steps 5 --warmup 3 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --one-stream > $GRAFT_REPO_ROOT/gpurun_out/rocprof_mixed.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/kstats_steady.py gpurun_out/prof_mixed/bench_kernel_trace.csv 4 200 > gpurun_out/r04_mixed_steady_v5.txt; head -30 gpurun_out/r04_mixed_steady_v5.txt | cut -c1-190
