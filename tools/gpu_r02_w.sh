#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_step.py tests/test_model_parity.py -q -x -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300
for i in 1 2 3; do
echo "prepared ahead"; timeout 900 python bench.py --no-cpu-baseline --no-fp32-leg 2>&1 | tail -1 | tee gpurun_out/bench_prep.json | cut -c100-200
echo "inside forward"; timeout 900 python bench.py --no-cpu-baseline --no-fp32-leg --no-prepare 2>&1 | tail -1 | cut -c100-200
done
