#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_mixed.py "tests/test_config_parity.py::test_gpu_config_fp32" tests/test_train_trajectory.py -q -m gpu -k "mixed or Mixed or second or winners or rounded" > gpurun_out/t_part.txt 2>&1; tail -8 gpurun_out/t_part.txt
for i in 1 2 3; do for e in "PCACC_UPCONV_CAT=0" "PCACC_UPCONV_CAT=1"; do
  ms=$(env $e timeout 900 python bench.py --dtype mixed --no-cpu-baseline --no-configs --no-fp32-leg --no-step-model 2>gpurun_out/err_mixed.txt | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
  echo "$e $ms"
done; done
