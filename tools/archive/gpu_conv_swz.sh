#!/bin/bash
# the 27-tap layers on swizzled LDS rows: isolated A/B, then the step with each kernel (interleaved), then the kernel's in-step duration
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_conv.py tests/test_step.py tests/test_model_parity.py -m gpu -q -x 2>&1 | tail -2
timeout 300 python tools/bench_conv_swz.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench_conv_swz.log
for i in 1 2 3; do
  for E in "PCACC_CONV_SWZ_OFF=1" "PCACC_X=0"; do
    ms=$(env $E timeout 900 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>&1 | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
    echo "$E $ms"
  done
done | tee gpurun_out/ab_conv_swz.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_swz -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model --one-stream > $GRAFT_REPO_ROOT/gpurun_out/rocprof_swz.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/kstats_steady.py gpurun_out/prof_swz/bench_kernel_trace.csv 4 200 > gpurun_out/swz_steady.txt; head -9 gpurun_out/swz_steady.txt; grep "conv3x3_resident_kernel" gpurun_out/swz_steady.txt | cut -c1-200
