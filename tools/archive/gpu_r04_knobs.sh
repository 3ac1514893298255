#!/bin/bash
# other runtime knobs, interleaved with the default (mixed, four sequences)
for i in 1 2 3 4; do for e in "PCACC_X=0" "HSA_ENABLE_INTERRUPT=0" "GPU_MAX_HW_QUEUES=8" "HSA_ENABLE_SDMA=0"; do
  out=$(env $e timeout 900 python bench.py --dtype mixed --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>gpurun_out/err_t.txt | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2))" 2>/dev/null)
  echo "$e $out"
done; done
