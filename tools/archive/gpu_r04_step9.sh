#!/bin/bash
# own bf16 transposed-convolution kernels: tests, A/B of the step (mixed + bf16), steady profiles
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_upconv_bf16.py tests/test_prepare_batch.py tests/test_conv_split.py -q -m gpu > gpurun_out/t_up.txt 2>&1; tail -12 gpurun_out/t_up.txt
timeout 2400 python -m pytest tests/test_mixed.py tests/test_model_parity.py tests/test_train_trajectory.py tests/test_step.py tests/test_config_parity.py -q -m gpu > gpurun_out/t_part.txt 2>&1; tail -8 gpurun_out/t_part.txt
for i in 1 2; do for e in "PCACC_UPCONV_BF16=0" "PCACC_UPCONV_BF16=1"; do for d in mixed bf16; do
  ms=$(env $e timeout 900 python bench.py --dtype $d --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>gpurun_out/err_mixed.txt | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$e $d B=4 $ms"
done; done; done
cd /tmp && export TMPDIR=/tmp
for d in mixed bf16; do
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$d -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --dtype $d --steps 5 --warmup 3 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --one-stream > $GRAFT_REPO_ROOT/gpurun_out/rocprof_$d.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kstats_steady.py $GRAFT_REPO_ROOT/gpurun_out/prof_$d/bench_kernel_trace.csv 4 200 > $GRAFT_REPO_ROOT/gpurun_out/r04_${d}_steady_v6.txt; head -12 $GRAFT_REPO_ROOT/gpurun_out/r04_${d}_steady_v6.txt | cut -c1-190
done
grep -n "upconv_bf16" $GRAFT_REPO_ROOT/gpurun_out/r04_mixed_steady_v6.txt $GRAFT_REPO_ROOT/gpurun_out/r04_bf16_steady_v6.txt | cut -c1-220
