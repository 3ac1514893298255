#!/bin/bash
# round 5: BatchNorm forward statistics pass with four rows in flight per lane (same summation order) against the previous library
python -m pytest tests/test_hip_ops.py tests/test_mixed.py -q -m gpu -x -k "bn or batch_norm or mixed" 2>&1 | tail -2
for i in 1 2; do
echo "== this tree"; python tools/bench_bn_ab.py 2>&1 | grep rows
echo "== previous commit"; PCACC_LIB=$PWD/build/libpcacc_hip_prev.so python tools/bench_bn_ab.py 2>&1 | grep rows
done
