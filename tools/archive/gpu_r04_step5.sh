#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_step.py tests/test_mixed.py -q -m gpu -k "collate or batcher or prepared or concatenation or mixed or rounded" > gpurun_out/t_part.txt 2>&1; tail -5 gpurun_out/t_part.txt
for i in 1 2 3; do for e in "PCACC_OWN_CAT=0" "PCACC_OWN_CAT=1"; do
  ms=$(env $e timeout 900 python bench.py --dtype mixed --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>gpurun_out/err_mixed.txt | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$e $ms"
done; done
bash tools/gpu_r04_pmc.sh
