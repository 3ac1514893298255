#!/bin/bash
# HIP runtime launch-path knob against the step's chains of small kernels: unset against HIP_FORCE_DEV_KERNARG=1, interleaved
for i in 1 2 3 4 5 6; do for e in "PCACC_X=0" "HIP_FORCE_DEV_KERNARG=1"; do
  out=$(env $e timeout 900 python bench.py --dtype mixed --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>gpurun_out/err_t.txt | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2))" 2>/dev/null)
  echo "$e $out"
done; done
