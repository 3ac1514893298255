#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3; do
echo "two streams"; timeout 900 python bench.py --no-cpu-baseline --no-fp32-leg 2>&1 | tail -1 | cut -c100-200
echo "one stream"; timeout 900 python bench.py --no-cpu-baseline --no-fp32-leg --one-stream 2>&1 | tail -1 | cut -c100-200
done
