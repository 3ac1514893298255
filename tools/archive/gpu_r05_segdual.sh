#!/bin/bash
# round 5: segment max with its bf16 copy from the same store (mixed mode) against PCACC_SEGMAX_DUAL=0 (the conversion pass of before), interleaved bench pairs
for i in 1 2 3; do
for sw in 1 0; do
  ms=$(PCACC_SEGMAX_DUAL=$sw timeout 900 python bench.py --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), round(d['ms_per_step_p50'],2))")
  echo "bench PCACC_SEGMAX_DUAL=$sw ms_per_step, p50: $ms"
done
done
