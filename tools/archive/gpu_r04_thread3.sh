#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3; do for e in "PCACC_EARLY_THREAD=0" "PCACC_X=auto"; do for b in 1 4; do for d in mixed bf16; do
  out=$(env $e timeout 900 python bench.py --dtype $d --batch $b --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>gpurun_out/err_t.txt | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), '|', d['config'].get('early_backward_thread'))" 2>/dev/null)
  echo "$e $d B=$b $out"
done; done; done; done
timeout 1500 python -m pytest tests/test_step.py tests/test_train_trajectory.py tests/test_bench_multirank.py -q -m gpu 2>&1 | tail -4
