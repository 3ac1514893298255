#!/bin/bash
# round 5: sorted bilinear backward with the first point of every tap segment fetched up front -- warm / cold A/B against the previous library, then the kernel in the step
python -m pytest tests/test_hip_ops.py -q -m gpu -x -k "bilinear or ungrid or gather" 2>&1 | tail -2
for i in 1; do
echo "== this tree"; python tools/bench_bilinear_ab.py 2>&1 | grep case
echo "== previous commit"; PCACC_LIB=$PWD/build/libpcacc_hip_prev.so python tools/bench_bilinear_ab.py 2>&1 | grep case
done
