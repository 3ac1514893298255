#!/bin/bash
# early backward issued from a helper thread: A/B at 1, 2, 4 sequences per step (mixed, bf16), then the tests of the step
mkdir -p gpurun_out
for i in 1 2; do for e in "PCACC_EARLY_THREAD=0" "PCACC_EARLY_THREAD=1"; do for b in 1 4; do for d in mixed bf16; do
  ms=$(env $e timeout 900 python bench.py --dtype $d --batch $b --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>gpurun_out/err_t.txt | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$e $d B=$b $ms"
done; done; done; done
timeout 1500 python -m pytest tests/test_step.py tests/test_train_trajectory.py tests/test_bench_multirank.py -q -m gpu 2>&1 | tail -4
