#!/bin/bash
for i in 1 2 3; do for e in "PCACC_PREFETCH_AT=forward" "PCACC_PREFETCH_AT=sync"; do for d in mixed bf16; do
  out=$(env $e timeout 900 python bench.py --dtype $d --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>gpurun_out/err_t.txt | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), d['roofline']['frac'])" 2>/dev/null)
  echo "$e $d $out"
done; done; done
